#!/bin/bash
# Round 6: the N > 1 code path at world size 1 over RCCL (the one GPU there is) — configs[3] and configs[4] at 100 M records, windows
# overlapped (default) and in sequence (--no-overlap) — and the host side of an 8-GPU node rehearsed with 8 processes (tools/host_8proc.py).
exec < /dev/null
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06w1; mkdir -p $O; cd $R
b() { name=$1; shift; timeout -k 5 400 python bench.py "$@" 2>$O/bench_$name.err | grep '^{' > $O/bench_$name.json; python -c "import json; j=json.load(open('$O/bench_$name.json')); print('$name', j['value'], j['ms_per_step'], j['roofline'].get('frac'), j['config'].get('exchange'))"; }
b world1_nccl_configs3_100m --gpus 1 --force-dist --backend nccl --records 100000000 --flows 1000000 --steps 5 --warmup 1
b world1_nccl_configs3_100m_no_overlap --gpus 1 --force-dist --backend nccl --records 100000000 --flows 1000000 --steps 5 --warmup 1 --no-overlap
b world1_nccl_configs4_100m --gpus 1 --force-dist --backend nccl --dedup --hot-permille 900 --records 100000000 --flows 1000000 --steps 5 --warmup 1
b world1_nccl_configs4_100m_no_overlap --gpus 1 --force-dist --backend nccl --dedup --hot-permille 900 --records 100000000 --flows 1000000 --steps 5 --warmup 1 --no-overlap
b world2_gloo_same_device_configs3 --gpus 2 --same-device --backend gloo --records 50000000 --flows 625000 --steps 3 --warmup 1
timeout 300 python tools/host_8proc.py --procs 8 --seconds 3 > $O/host_8proc.txt 2>&1; tail -12 $O/host_8proc.txt
timeout 300 python tools/host_8proc.py --procs 8 --seconds 3 --unbound > $O/host_8proc_unbound.txt 2>&1; tail -4 $O/host_8proc_unbound.txt
