#!/usr/bin/env python3
"""What the link gives nfagg_account's host pipeline: 8 M records up (1.15 GB) and 2.78 M evicted flows down (0.40 GB), page-locked
buffers, each direction alone and both at once (chunks of 1 Mi records on two streams, as the pipeline issues them)."""
import time
import torch
n_up, n_dn, chunks = 8_000_000 * 144, 2_782_941 * 144, 8
h_up = torch.empty(n_up, dtype=torch.uint8).pin_memory(); d_up = torch.empty(n_up, dtype=torch.uint8, device="cuda")
h_dn = torch.empty(n_dn, dtype=torch.uint8).pin_memory(); d_dn = torch.empty(n_dn, dtype=torch.uint8, device="cuda")
s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
cu, cd = n_up // chunks, n_dn // chunks


def run(up, dn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(chunks):
        if up:
            with torch.cuda.stream(s_up):
                d_up[k * cu:(k + 1) * cu].copy_(h_up[k * cu:(k + 1) * cu], non_blocking=True)
        if dn:
            with torch.cuda.stream(s_dn):
                h_dn[k * cd:(k + 1) * cd].copy_(d_dn[k * cd:(k + 1) * cd], non_blocking=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for name, up, dn in (("up alone", 1, 0), ("down alone", 0, 1), ("both at once", 1, 1)):
    run(up, dn)
    ts = [run(up, dn) for _ in range(5)]
    t = min(ts)
    print(f"{name:14s} {t * 1e3:7.2f} ms  up {n_up * up / t / 1e9:6.1f} GB/s  down {n_dn * dn / t / 1e9:6.1f} GB/s   (all: {[round(x * 1e3, 2) for x in ts]})")
