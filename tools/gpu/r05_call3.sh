#!/bin/bash
# round 5: the host copy workers (pool bound to the GPU's NUMA node, non-temporal copies, calibrated parts): ring drain, host paths
cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/r05c3; mkdir -p $O
export TMPDIR=/tmp
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)" | head -8 > $O/host.txt; cat $O/host.txt
timeout 300 python -m pytest tests/test_ring_to_device_gpu.py tests/test_c_driver.py tests/test_account_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -4 | tee $O/tests.txt
for mode in pinned pageable; do timeout 120 python tools/ring_drain_bench.py $mode 2>&1 | grep -v amdgpu | tail -6; done | tee $O/ring_drain.txt
timeout 200 python tools/account_paths_bench.py --variant 0 --reps 3 2>&1 | grep -v amdgpu | tail -1 | tee $O/paths.txt
timeout 200 python tools/host_path_bench.py 2>&1 | grep -v amdgpu | tail -12 | tee $O/host_path.txt
