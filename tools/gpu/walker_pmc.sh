#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$PWD/gpurun_out/walker_pmc; rm -rf $O; mkdir -p $O
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_BRANCH"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-30)
  (cd /tmp && timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/$tag -o t -- python $GRAFT_REPO_ROOT/tools/account_5000_prof.py --steps 1 > /dev/null 2> $O/$tag.err)
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0]
    if "k_par_cuts" in k or "k_par_segfold" in k or "k_par_links" in k: agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items(): print(k[:40], dict(v))
PY
done
