#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03y; mkdir -p $OUT
timeout -k 5 300 python bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 > $OUT/b1.json 2> $OUT/b1.err; echo "rc $? lines $(wc -l < $OUT/b1.json)"; python -c "import json; j=json.load(open('$OUT/b1.json')); print(j['value'])"
timeout -k 5 300 python bench.py --gpus 2 --same-device --backend gloo --records 10000000 --flows 300000 --steps 2 --warmup 1 > $OUT/b2.json 2> $OUT/b2.err; echo "rc $? lines $(wc -l < $OUT/b2.json)"; python -c "import json; j=json.load(open('$OUT/b2.json')); print(j['value'], j['n_gpus'])"
timeout -k 5 300 python bench.py --group-devices 0 --records 10000000 --flows 300000 --sketches --steps 2 --warmup 1 > $OUT/b3.json 2> $OUT/b3.err; echo "rc $? lines $(wc -l < $OUT/b3.json)"; python -c "import json; j=json.load(open('$OUT/b3.json')); print(j['value'])"; grep -c "RCCL version" $OUT/b3.err
timeout -k 5 300 python -m pytest tests/test_partials_gpu.py -x -q -m gpu -k "rehearsal or bench" 2>&1 | tail -2
