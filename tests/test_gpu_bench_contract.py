"""bench.py's output contract (the driver parses ONE JSON line): required keys, types and the roofline /
cpu_baseline objects, on a small workload so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, legs=False):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--particles", "300000"] + ([] if legs else ["--no-legs"]) + list(extra),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout          # exactly one line on stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contracts_fields():
    d = run_bench("--traffic", "off")
    assert d["metric"] == "snapshot pack+write GB/s at 10M particles/GPU" and d["unit"] == "GB/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert isinstance(d["value"], float) and d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["target_dir"] == "/dev/shm" and d["config"]["target_fstype"] == "tmpfs"
    assert d["comm_backend"] == "self" and d["collectives_per_frame"] == 0.0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes_per_launch"] == 56 * 300000 and r["kernel"] == "pack_rows_kernel"
    assert r["traffic"] is None                     # --traffic off
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["unit"] == "GB/s" and c["value"] > 0
    assert isinstance(c["sample"], str) and "particles" in c["sample"]
    if c["kind"] == "reference":                    # ranks pinned, and the line says how
        assert "bind-to" in c["sample"] and "bind-to" in c["binding"]     # "user:<cores of the GPU's NUMA node>" or "core"
    assert "legs" not in d                          # --no-legs
    # what a draining snapshot costs a kernel that runs beside it (the copies are shader blits): measured, not assumed
    o = d["snapshot_overlap"]
    assert isinstance(d["snapshot_overlap_slowdown_pct"], float) and d["snapshot_overlap_slowdown_pct"] == o["slowdown_pct"]
    assert o["launches"] >= 20 and o["ms_per_launch_alone"] > 0 and o["frames_drained"] == 4
    assert d["snapshot_stall_ms"] > 0


def test_bench_measures_its_hbm_traffic_with_the_pmc_counters():
    """roofline.traffic comes from two rocprofv3 counter passes of the run itself: read bytes = FETCH_SIZE x 2
    (gfx950), written bytes = WRITE_SIZE; for the row kernel they are the source arrays and the chunks."""
    d = run_bench("--no-cpu-baseline")
    r = d["roofline"]
    if r["traffic"] is None:
        pytest.skip("rocprofv3 not usable on this box: " + str(r["traffic_source"]))
    assert r["traffic_source"].startswith("live")
    n = 300000
    det = r["traffic_detail"]
    assert abs(det["write_bytes"] - 28 * n) < 0.02 * 28 * n          # the three chunks
    assert abs(det["read_bytes"] - 32 * n) < 0.05 * 32 * n           # two float4 arrays
    assert r["traffic"] == det["read_bytes"] + det["write_bytes"]


def test_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself.  The box has one GPU: without the
    rehearsal flag that is refused (one rank per GPU is the contract), with it both ranks share cuda:0 over a gloo
    group and the line says n_gpus 2, one collective per frame, and what that collective cost."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
            "--particles", "200000", "--traffic", "off"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() < 2:
        p = subprocess.run(base, capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode != 0 and p.stdout.strip() == ""
        assert "this node shows 1 GPU" in p.stderr and "--rehearse-shared-gpu" in p.stderr
    p = subprocess.run(base + ["--rehearse-shared-gpu"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collectives_per_frame"] == 1.0 and d["comm_backend"] == "torch-gloo"
    x = d["exchange_us"]
    assert x["per_rank_count"] == 3 and 0 < x["min"] <= x["mean_fastest_rank"] <= x["mean"] <= x["max"]
    assert x["bytes_per_rank"] == 512
    c = d["cpu_baseline"]
    assert c["nproc"] >= 1 and c["value"] > 0
    if c["kind"] == "reference":                    # P host ranks for P GPUs; the port is one thread
        assert c["ranks"] == 2 and c["cores"] == 2 and c["more_cores"]["ranks"] >= 2
    assert d["config"]["parallelism"] == "particle-partition x2"


def test_under_torch_distributed_run_the_way_the_driver_starts_it():
    """The driver's own command line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` -- with two ranks sharing the one GPU of the box:
    rank 0 prints exactly ONE line on stdout, the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)
    is what the ranks rendezvous on, and a WORLD_SIZE that disagrees with --gpus is refused by every rank."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")]
    args = ["--steps", "3", "--warmup", "1", "--particles", "200000", "--traffic", "off", "--no-cpu-baseline",
            "--rehearse-shared-gpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run(launch + ["--gpus", "2"] + args, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["collectives_per_frame"] == 1.0 and d["config"]["parallelism"] == "particle-partition x2"
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and "legs" not in d          # legs: one rank only
    p = subprocess.run(launch + ["--gpus", "4"] + args, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in p.stderr


def test_single_rank_line_carries_the_exchange_probe():
    d = run_bench("--traffic", "off", "--no-cpu-baseline")
    assert d["exchange_us"] is None                 # one rank: no exchange
    pr = d["exchange_probe"]
    if pr is None:
        pytest.skip("no RCCL on this box")
    assert pr["bytes"] == 512 and 0 < pr["min_us"] <= pr["median_us"] <= pr["p99_us"]


def test_declared_partition_run_costs_no_collective_per_frame():
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
            "--particles", "200000", "--traffic", "off", "--no-cpu-baseline", "--rehearse-shared-gpu", "--declared-partition"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run(base, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 2 and d["collectives_per_frame"] == 0.0 and "declared" in d["placement"]
    assert d["exchange_us"]["per_rank_count"] == 0 and d["value"] > 0


def test_a_target_without_room_stops_the_run_instead_of_moving_it():
    """when --dir lacks room rank 0 used to move the run to /tmp or the working directory
    silently, so a 1 -> 8 curve could mix file systems.  Now the run stops with the figures unless --dir-fallback is
    given (and names directory and file system type in the line either way)."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "400000", "--warmup", "1",
            "--particles", "10000000", "--traffic", "off", "--no-cpu-baseline"]        # 112 TB of frames
    p = subprocess.run(base, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "has not that much room" in p.stderr and "--dir-fallback" in p.stderr and "nor /tmp" not in p.stderr
    p = subprocess.run(base + ["--dir-fallback"], capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == "" and "nor /tmp" in p.stderr


LEG_NAMES = ["config2", "config4_sph", "config4_union", "config5_read", "gather_uniform", "gather_hilbert"]


def test_bench_line_carries_a_leg_for_every_other_baseline_config():
    """configs 2, 4 (sph, union), 5 and the tag-order gather sit in the driver-run line, each with
    avg_us / algorithmic_bytes / frac (the reference's harnesses print every leg they run, benchmark-write.cc:144-172,
    benchmark-read.cc:128-146).  Small particle count here; config 2 keeps its own 2^20 rows."""
    d = run_bench("--traffic", "off", "--no-cpu-baseline", "--no-stall-test", "--no-exchange-probe", legs=True)
    legs = {e["name"]: e for e in d["legs"]}
    assert list(legs) == LEG_NAMES
    n = 300000
    for name, e in legs.items():
        assert "error" not in e, e
        assert e["avg_us"] > 0 and e["algorithmic_bytes"] > 0 and 0 < e["frac"] < 1.0, e
        assert abs(e["frac"] - e["algorithmic_bytes"] / (e["avg_us"] * 1e-6) / 8e12) < 2e-3
        assert e["traffic"] is None                 # --traffic off
    c2 = legs["config2"]
    assert c2["algorithmic_bytes"] == 56 * (1 << 20) and c2["launches"] == 200 and c2["buffer_sets"] == 11
    assert c2["unrotated_us"] > 0 and c2["hoomd_layout_us"] > 0 and c2["double4_us"] > 0 and c2["target_frac"] == 0.70
    assert legs["config4_sph"]["algorithmic_bytes"] == 224 * n and legs["config4_union"]["algorithmic_bytes"] == 328 * n
    for k in ("config4_sph", "config4_union"):
        assert legs[k]["frames"] == 3 and legs[k]["value_GBps"] > 0 and legs[k]["target_fstype"] == "tmpfs"
    c5 = legs["config5_read"]
    assert c5["bit_exact"] is True and c5["file_to_hbm_GBps"] > 0 and c5["algorithmic_bytes"] == 56 * n
    assert c5["file_rows"] == 8 * n and c5["rows"] == n
    assert legs["gather_uniform"]["algorithmic_bytes"] == 60 * n == legs["gather_hilbert"]["algorithmic_bytes"]
    # the essentials again as flat scalars of `roofline`: what the driver's record keeps of the line
    r = d["roofline"]
    for name, e in legs.items():
        assert r["legs_%s_us" % name] == e["avg_us"] and r["legs_%s_frac" % name] == e["frac"]
    assert all(not isinstance(v, (dict, list)) for k, v in r.items() if k.startswith("legs_"))
    assert d["legs_wall_s"] > 0


def test_legs_measure_their_hbm_traffic():
    """One FETCH_SIZE and one WRITE_SIZE pass over a child that launches every leg's kernels, the legs told apart by a
    marker launch: the streaming legs move about their algorithmic bytes, the random gather moves several times its."""
    d = run_bench("--no-cpu-baseline", "--no-stall-test", "--no-exchange-probe", legs=True)
    if d["roofline"]["traffic"] is None:
        pytest.skip("rocprofv3 not usable on this box")
    legs = {e["name"]: e for e in d["legs"]}
    assert d["legs_traffic_source"].startswith("live")
    for k in ("config2", "config4_sph", "config4_union", "config5_read"):
        e = legs[k]
        assert e["traffic"] is not None and e["fetch_x2"] is True, e
        assert 0.9 < e["traffic_over_algorithmic"] < 1.35, e
        assert d["roofline"]["legs_%s_traffic_ratio" % k] == e["traffic_over_algorithmic"]
    assert legs["gather_uniform"]["fetch_x2"] is False
    assert legs["gather_uniform"]["traffic_over_algorithmic"] > legs["gather_hilbert"]["traffic_over_algorithmic"] * 0.9
