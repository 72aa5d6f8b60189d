#!/bin/bash
# round 3: nfagg_account (persistent epoch kernel) — parity, then the small-table legs of the bench
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03d; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_account_gpu.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -25 $OUT/pytest.txt
timeout -k 5 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc $?"; python - <<'PY'
import json
j = json.load(open("gpurun_out/r03d/bench.json"))
print(j["value"], j["ms_per_step"])
print(json.dumps(j.get("extra", {}).get("cache_max_flows_5000"), indent=1))
print(json.dumps(j.get("extra", {}).get("e2e"), indent=1))
PY
tail -3 $OUT/bench.err
