import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth
n, keys = 50_000_000, 1_000_000
th = synth.zipf_thresholds(keys, 1.1); d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
synth.stream_device(d.data_ptr(), n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr(), hot_permille=900, variant=2); torch.cuda.synchronize()
out = torch.empty((keys + 16) * 144, dtype=torch.uint8, device="cuda")
with nf.FlowTable(max_entries=1 << 21, mode=nf.MODE_KERNEL_DEDUP, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, profile=True) as tab:
    for rep in range(3):
        tab.ingest_device(d.data_ptr(), n); tab.evict_device(out.data_ptr(), keys + 16); tab.sketch_reset()
    st = tab.stats()
    print(f"dedup + sketches, 90 % hot: k_sketch_update {st.sketch_kernel_ms / max(1, st.sketch_launches):.3f} ms per {n} records, fold {st.ingest_kernel_ms / max(1, st.ingest_launches):.3f} ms")
