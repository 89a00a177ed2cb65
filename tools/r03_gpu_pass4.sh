#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_pass4
mkdir -p $O
timeout -k 10 300 python tools/pwrite_source_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/pwrite_source_probe.log
echo "--- pinned to the GPU's node (cpu 0-7)"
timeout -k 10 300 taskset -c 0-7 python tools/pwrite_source_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/pwrite_source_probe.log
