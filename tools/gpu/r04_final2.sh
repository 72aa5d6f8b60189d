#!/bin/bash
# after the ring-drain change: the ring tests, the C driver, and the driver's default bench line again
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04z2
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ring_to_device_gpu.py tests/test_c_driver.py tests/test_ringbuf.py tests/test_parity_gpu.py -q -m gpu 2>&1 | tail -3
timeout 500 python bench.py > gpurun_out/r04z2/bench_n1.json 2> gpurun_out/r04z2/bench_n1.err
echo "bench rc $?"; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04z2/bench_n1.json') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['frac'], j['roofline']['frac_traffic'])
for k in ('e2e','e2e_page_locked','e2e_ring'):
    print(k, json.dumps({a:b for a,b in j['extra'][k].items() if a not in ('what','bound')}))
PY
