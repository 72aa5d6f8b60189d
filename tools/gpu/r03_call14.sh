#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_account_gpu.py tests/test_parity_gpu.py tests/test_c_driver.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $OUT/pytest.txt
timeout -k 5 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03m/bench.json"))
print(j["value"], j["ms_per_step"])
ex = j.get("extra", {})
print({k: v for k, v in ex.get("cache_max_flows_5000", {}).items() if k != "what"})
print(ex.get("e2e"))
PY
