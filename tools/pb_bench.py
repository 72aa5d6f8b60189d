#!/usr/bin/env python3
"""Throughput of the device-resident evict -> protobuf encode hand-off (run on the GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth

flows, n = 1_000_000, 20_000_000
th = synth.zipf_thresholds(flows, 1.1)
d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
synth.stream_device(d.data_ptr(), n, seed=2, n_keys=flows, d_thresholds=d_th.data_ptr())
torch.cuda.synchronize()
names = nf.intf_table([(2, None, "eth0", ""), (3, None, "eth1", "default"), (4, None, "br-ex", ""), (5, None, "ovn-k8s-mp0", "blue")])
agent = bytes(10) + b"\xff\xff" + bytes([10, 0, 0, 1])
with nf.FlowTable(max_entries=1 << 22) as tab:
    tab.ingest_device(d.data_ptr(), n)
    d_ev = torch.empty(flows * 144 + 16, dtype=torch.uint8, device="cuda")
    m = tab.evict_device(d_ev.data_ptr(), flows)
    d_off = torch.empty(m + 1, dtype=torch.int64, device="cuda"); d_len = torch.empty(m, dtype=torch.int32, device="cuda")
    d_keys = torch.empty(m * 32, dtype=torch.uint8, device="cuda")
    rc, need = tab.encode_pb_device(d_ev.data_ptr(), m, 10**18, 10**12, agent, names, 0, 0, d_off.data_ptr(), d_len.data_ptr())
    d_out = torch.empty(need + 16, dtype=torch.uint8, device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, wrote = tab.encode_pb_device(d_ev.data_ptr(), m, 10**18, 10**12, agent, names, d_out.data_ptr(), need, d_off.data_ptr(), d_len.data_ptr(), d_keys.data_ptr())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{m} flows -> {wrote} bytes ({wrote / m:.1f} B/record) in {dt * 1e3:.3f} ms = {m / dt / 1e6:.1f} M records/s, {(m * 144 + wrote) / dt / 1e9:.1f} GB/s read+written")
