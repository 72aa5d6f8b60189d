#!/usr/bin/env python3
"""Test infrastructure (under tests/: the CPU oracle is timed beside the device path). Throughput of the device-resident LookupAndDeleteMap join (nfagg_map_merge_device) and of the
merged-flows -> protobuf hand-off, against the CPU oracle's join on the same arrays (run on the GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import netobserv_ebpf_agent_amd as nf
from oracle import oracle as O

n_main, n_feat, n_cpu = 1_000_000, 400_000, 16
rng = np.random.default_rng(5)
pop = O.gen_stream(n_main + n_feat // 2, seed=3, n_keys=1 << 40)["id"].copy()       # distinct ids
pop = pop[np.unique(pop.view("V40"), return_index=True)[1]]
mi = pop[:n_main].copy(); mv = np.zeros(len(mi), dtype=O.FLOW_METRICS)
mv.view(np.uint8).reshape(len(mi), 104)[:] = rng.integers(0, 256, (len(mi), 104), dtype=np.uint8)
feats = {}
for kind in ("dns", "drops", "xlat", "additional"):
    dt = O.KIND_DTYPES[O.KIND_INDEX[kind]]
    fi = pop[rng.permutation(len(pop))[:n_feat]].copy()
    fv = np.zeros((len(fi), n_cpu), dtype=dt)
    fv.view(np.uint8).reshape(len(fi), -1)[:] = rng.integers(0, 256, (len(fi), n_cpu * dt.itemsize), dtype=np.uint8)
    feats[kind] = (fi, fv)
rows = len(mi) + sum(len(a) for a, _ in feats.values())
in_bytes = mi.nbytes + mv.nbytes + sum(a.nbytes + b.nbytes for a, b in feats.values())
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
d_mi, d_mv = dev(mi), dev(mv)
d_f = {k: (dev(a), dev(b), len(a)) for k, (a, b) in feats.items()}
sizes = {"records": 144, "present": 1, "additional": 32, "dns": 64, "drops": 32, "network_events": 72, "xlat": 56, "quic": 24}
d_out = {k: torch.zeros(rows * s + 16, dtype=torch.uint8, device="cuda") for k, s in sizes.items()}
names = nf.intf_table([(2, None, "eth0", ""), (3, None, "eth1", "default")])
agent = bytes(10) + b"\xff\xff" + bytes([10, 0, 0, 1])
with nf.FlowTable(max_entries=64) as tab:
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, n, dups = tab.map_merge_device((d_mi.data_ptr(), d_mv.data_ptr(), len(mi)), {k: (i.data_ptr(), v.data_ptr(), m) for k, (i, v, m) in d_f.items()},
                                           n_cpu, {k: t.data_ptr() for k, t in d_out.items()}, rows)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"nfagg_map_merge_device: {rows} rows ({in_bytes / 1e9:.2f} GB of drained maps, {n_cpu} CPUs) -> {n} flows in {dt * 1e3:.3f} ms = "
          f"{n / dt / 1e6:.1f} M flows/s, {in_bytes / dt / 1e9:.0f} GB/s of input")
    d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
    ptrs = {k: d_out[k].data_ptr() for k in ("additional", "dns", "drops", "xlat", "quic")}
    rc, need = tab.encode_pb_device(d_out["records"].data_ptr(), n, 10**18, 10**12, agent, names, 0, 0, d_off.data_ptr(), d_len.data_ptr(),
                                    d_present=d_out["present"].data_ptr(), d_parts=ptrs)
    d_pb = torch.empty(need + 16, dtype=torch.uint8, device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, wrote = tab.encode_pb_device(d_out["records"].data_ptr(), n, 10**18, 10**12, agent, names, d_pb.data_ptr(), need, d_off.data_ptr(), d_len.data_ptr(),
                                         d_present=d_out["present"].data_ptr(), d_parts=ptrs)
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print(f"nfagg_encode_pb_content_device: {n} flows -> {wrote} bytes ({wrote / n:.1f} B/flow) in {dt2 * 1e3:.3f} ms = {n / dt2 / 1e6:.1f} M flows/s")
t0 = time.perf_counter()
ids, contents = O.map_merge(mi, mv, feats, n_cpu)
cpu = time.perf_counter() - t0
print(f"CPU oracle join (1 core, C restatement of LookupAndDeleteMap): {len(ids)} flows in {cpu * 1e3:.0f} ms = {len(ids) / cpu / 1e6:.2f} M flows/s")
