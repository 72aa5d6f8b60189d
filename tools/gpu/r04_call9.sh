#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04e
export TMPDIR=/tmp
R=$PWD
( MAX_ENTRIES=16777216 timeout 200 python tools/phase_timing.py 10000000 100000000 8; MAX_ENTRIES=16777216 timeout 200 python tools/phase_timing.py 10000000 100000000 9 ) 2>&1 | grep -v amdgpu > gpurun_out/r04e/phase_10m.txt
cat gpurun_out/r04e/phase_10m.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04e/trace10m -- python $R/bench.py --flows 10000000 --max-entries 16777216 --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $R/gpurun_out/r04e/bench_10m.json 2> $R/gpurun_out/r04e/bench_10m.err
cd $R
f=$(find gpurun_out/r04e/trace10m -name "*kernel_stats.csv" | head -1); head -12 $f
tail -c 1500 gpurun_out/r04e/bench_10m.json
