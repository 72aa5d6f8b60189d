#!/bin/bash
# round 5 kept experiment: a wave's duplicates combined before the LDS atomics of pass 1 (libnfagg_diag.so ingest_variant 24) —
# parity, time of the 100 M-record fold call against the shipping kernel on the same box, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of both
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$PWD/gpurun_out/wave_combining; rm -rf $O; mkdir -p $O
export NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so
timeout 200 python tests/tools/diag_variant_parity.py 24 2>&1 | grep -v amdgpu | tail -5 | tee $O/parity.txt
for v in 10 24 10 24; do
  timeout 200 python bench.py --variant $v --no-extras --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | grep '^{' > $O/bench_$v.json
  python3 -c "import json; j=json.load(open('$O/bench_$v.json')); print('variant $v: fold call', j['roofline']['launch_ms'], 'ms,', j['value'], 'M records/s')" | tee -a $O/times.txt
done
for v in 10 24; do
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_$v -- python $GRAFT_REPO_ROOT/bench.py --variant $v --no-extras --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> $O/pmc_$v.err)
  f=$(find $O/pmc_$v -name "*counter_collection.csv" | head -1)
  python3 - "$f" $v <<'PY' | tee -a $O/counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pass1" in r["Kernel_Name"]: acc[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print("variant", sys.argv[2], k, {n: int(v) for n, v in c.items()}, "conflict/active = %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"])))
PY
done
