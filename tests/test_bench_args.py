"""bench.py's launch contract, the part that needs no GPU: `--gpus N` IS the number of ranks.

Round 2's bench parsed `--gpus` and never read it: `python bench.py --gpus 8` without a launcher ran one
rank and printed n_gpus 1 (missing #1).  Now the flag either matches the launcher's WORLD_SIZE or
makes the script launch N ranks itself; anything else exits non-zero with a message."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_world_size_must_equal_gpus():
    p = run(["--gpus", "4", "--steps", "1", "--warmup", "0"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "WORLD_SIZE=2" in p.stderr and "--gpus 4" in p.stderr
    p = run(["--steps", "1"], WORLD_SIZE="2", RANK="1", LOCAL_RANK="1")       # default --gpus 1 under a 2-rank launcher
    assert p.returncode != 0 and "must agree" in p.stderr


def test_gpus_below_one_is_refused():
    p = run(["--gpus", "0"])
    assert p.returncode != 0 and "at least 1" in p.stderr


def test_self_launch_starts_n_ranks_and_reports_their_failure():
    """No launcher, --gpus 2: the parent starts two rank children (RANK 0 and 1, WORLD_SIZE 2) before touching
    torch.  On this GPU-less container each child refuses to run (no CPU fallback) and the parent exits
    non-zero instead of printing a one-rank line."""
    import torch
    if torch.cuda.is_available():
        return      # the GPU-side launch is tests/test_gpu_bench_contract.py's
    p = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--particles", "1000"])
    assert p.returncode != 0
    assert p.stdout.strip() == ""                               # no JSON line claims a run that did not happen
    assert p.stderr.count("needs an MI355X") == 2               # both ranks were started and said why


def test_target_file_system_is_named():
    """`config.target_fstype` comes from /proc/mounts (longest mount point that holds the directory): a 1 -> 8 curve
    whose points landed on different file systems shows it."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.fstype_of("/proc/self") == "proc"
    if os.path.isdir("/dev/shm"):
        assert bench.fstype_of("/dev/shm") == "tmpfs"
    assert bench.fstype_of(ROOT) not in (None, "proc", "sysfs")
    assert bench.fstype_of("/no/such/dir/anywhere") is not None           # falls to the root's


def test_ranks_get_the_ipc_mode_the_pool_needs():
    """bench.py exports HSA_ENABLE_IPC_MODE_LEGACY=0 to itself (setdefault) before torch / HIP load, for self-launched
    and launcher-started ranks alike: the hosts only support dmabuf IPC handles, without which RCCL's intra-node
    transport fails in hipIpcGetMemHandle."""
    src = open(BENCH).read()
    at_set = src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")')
    assert at_set < src.index("import torch\n", src.index("def main()"))
    assert 'HSA_ENABLE_IPC_MODE_LEGACY="0"' not in src                     # no longer only in the self-launch path
