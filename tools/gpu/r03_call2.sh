#!/bin/bash
# round 3: where does pass 1's time go? Ablated builds of k_pass1 (libnfagg_diag.so, wrong results, timing only)
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $OUT
export NFAGG_LIB=$GRAFT_REPO_ROOT/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so
cd /tmp
for v in 10 21 22 23 25 27; do
  timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/abl_$v -o abl -- python $GRAFT_REPO_ROOT/bench.py --variant $v --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/abl_$v.json 2> $OUT/abl_$v.err
  f=$(find $OUT/abl_$v -name '*kernel_stats.csv' | head -1)
  echo "== variant $v ($f)"
  [ -n "$f" ] && grep -E "k_pass1|k_pass2|k_finalize|k_evict" "$f" | cut -c1-220
done
