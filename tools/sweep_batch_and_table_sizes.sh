R=$PWD
echo "chunk variant Mrec/s launch_ms(per call)"
for chunk in 65536 262144 1048576 2097152 3145728 4194304; do
 for v in 7 10; do
  timeout 120 python bench.py --records 25165824 --flows 1000000 --chunk $chunk --variant $v --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print($chunk, $v, j['roofline']['kernel_Mrecords_per_s'], j['roofline']['launch_ms'])"
 done
done
echo "max_entries value ms_per_step evicted/step"
for me in 5000 100000 524288; do
  timeout 200 python - $me <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth
me = int(sys.argv[1]); n, keys = 20_000_000, 1_000_000
th = synth.zipf_thresholds(keys, 1.1); d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
synth.stream_device(d.data_ptr(), n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr()); torch.cuda.synchronize()
out = torch.empty((me + 16) * 144, dtype=torch.uint8, device="cuda")
with nf.FlowTable(max_entries=me) as tab:
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); off = 0; ev = 0; fl = 0
        while off < n:
            rc, c = tab.ingest_device(d.data_ptr() + off * 144, n - off); off += c
            if rc == nf.FULL:
                fl += tab.evict_device(out.data_ptr(), me + 16, nf.REASON_FULL); ev += 1
        fl += tab.evict_device(out.data_ptr(), me + 16, nf.REASON_CLOSING); ev += 1
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = tab.stats()
    print(f"max_entries {me}: {n / dt / 1e6:.1f} M records/s, {ev} evictions ({fl} flows) per 20 M records, optimistic folds {st.optimistic_folds} rollbacks {st.optimistic_rollbacks}")
PY
done
