#!/bin/bash
exec < /dev/null
# Collect the rocprofv3 evidence for bench.py on the GPU box (run from the repo root via gpurun).
# 1) kernel-trace + stats  2) PMC passes (each in its own run, no other trace domains)
# 3) the same FETCH_SIZE / WRITE_SIZE passes over tools/pmc_calib (known byte counts) for calibration.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --cpu-sample 0 --no-extras}"
PMC_ARGS="${PMC_BENCH_ARGS:---steps 1 --warmup 0 --cpu-sample 0 --no-extras}"
cd /tmp
PROG="${PROF_PROG:-python $R/bench.py}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $PROG $ARGS > $OUT/trace_bench.json 2> $OUT/trace_err.txt
# PMC_LIGHT=1: only the two traffic counters (the extra legs of round 4: one FETCH_SIZE and one WRITE_SIZE pass each)
if [ "${PMC_LIGHT:-0}" = "1" ]; then GROUPS_=("FETCH_SIZE" "WRITE_SIZE"); else GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"); fi
PROG="${PROF_PROG:-python $R/bench.py}"
for c in "${GROUPS_[@]}"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$tag -- $PROG $PMC_ARGS > $OUT/pmc_${tag}_bench.json 2> $OUT/pmc_${tag}_err.txt
done
if [ -x $R/tools/pmc_calib ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$c -- $R/tools/pmc_calib > $OUT/calib_${c}_out.txt 2> $OUT/calib_${c}_err.txt
  done
fi
cd $R
# what was profiled: the library this box ran (bench.py marks a traffic figure stale when the library it loads is another one)
sha256sum $R/netobserv-ebpf-agent_amd/lib/libnfagg.so | cut -d' ' -f1 > $OUT/lib_sha256.txt
find $OUT -name "*.csv" | wc -l
