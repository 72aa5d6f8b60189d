#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$PWD/gpurun_out/gather_flavours; rm -rf $O; mkdir -p $O
timeout 120 tools/gpu/gather_flavours | tee $O/times.txt
(cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc -- $GRAFT_REPO_ROOT/tools/gpu/gather_flavours > /dev/null 2> $O/pmc_err.txt)
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/fetch.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    kib = sum(v) / len(v)
    print(f"{k[:60]:60s} launches {len(v)}  FETCH_SIZE {kib:12.0f} KiB  x2 -> {kib * 2 * 1024 / 35e6:6.1f} B per gathered record")
PY
