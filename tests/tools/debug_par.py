#!/usr/bin/env python3
"""Test infrastructure: GPU debugging aid (libnfagg_diag.so): the cuts the epochs-found-first path of nfagg_account found against the prefix-count rule
computed on the CPU (tests/test_epoch_boundaries.py)."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/tools -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("NFAGG_LIB", os.path.join(ROOT, "netobserv-ebpf-agent_amd", "lib", "libnfagg_diag.so"))
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import _lib as L
from oracle import oracle as O
from test_epoch_boundaries import prev_links, epoch_cuts, key_ids

L.lib.nfagg_debug_last_cuts.restype = C.c_int
L.lib.nfagg_debug_last_cuts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t]


def last(tab, cap=70000):
    ctl = (C.c_uint32 * 8)(); cuts = (C.c_uint32 * cap)()
    assert L.lib.nfagg_debug_last_cuts(tab._h, ctl, cuts, cap) == 0
    return list(ctl), np.frombuffer(cuts, dtype=np.uint32).copy()


L.lib.nfagg_debug_last_analysis.restype = C.c_int
L.lib.nfagg_debug_last_analysis.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]


def analysis(tab, n):
    ks = np.zeros(n, dtype=np.uint64); prev = np.zeros(n, dtype=np.int32); pos = np.zeros(n, dtype=np.uint32)
    assert L.lib.nfagg_debug_last_analysis(tab._h, ks.ctypes.data_as(C.c_void_p), prev.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), n) == 0
    return ks, prev, pos


def case(n, keys, M, seed, variant, device):
    import torch
    recs = O.gen_stream(n, seed=seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), variant=variant)
    want = epoch_cuts(prev_links(key_ids(recs)), M)
    with nf.FlowTable(max_entries=M) as tab:
        err = None
        try:
            if device:
                d = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).cuda()
                out = torch.zeros((n + M) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
                rc, c, ends = tab.account_device(d.data_ptr(), n, out.data_ptr(), n + M, n // M + 16)
            else:
                rc, c, ep = tab.account(recs.view(nf.FLOW_RECORD))
        except Exception as e:
            err = str(e)[:160]
        ctl, cuts = last(tab)
        ks, prev, pos = analysis(tab, n)
    wp = prev_links(key_ids(recs))
    bad = np.nonzero(prev.astype(np.int64) != wp)[0]
    srt = bool((np.diff(ks.astype(np.uint64)) > 0).all()) if n > 1 else True
    idx_ok = bool((np.sort((ks & np.uint64(0xFFFFFFFF)).astype(np.int64)) == np.arange(n)).all())
    print("   sorted strictly ascending:", srt, "indices a permutation:", idx_ok, "prev mismatches:", len(bad), "first:", bad[:5].tolist(),
          [(int(prev[b]), int(wp[b])) for b in bad[:5]], flush=True)
    hb = np.zeros(n, dtype=np.uint64); hb[(ks & np.uint64(0xFFFFFFFF)).astype(np.int64)] = ks >> np.uint64(32)
    raw = recs.view(np.uint8).reshape(n, 144)
    def cpu_hb(i):
        k = raw[i, :40].copy(); k[39] = 0
        return O.lib().orc_key_hash(k.ctypes.data_as(C.c_void_p)) >> 32
    ok_wants = wp[bad]
    diff_hb = int((hb[bad] != hb[ok_wants]).sum())
    print("   mismatches whose two records got different hash bits on the device:", diff_hb, "of", len(bad))
    for b in bad[:3]:
        w_ = int(wp[b])
        print("   idx", int(b), "dev hb", hex(int(hb[b])), "cpu", hex(cpu_hb(int(b))), "| prev", w_, "dev hb", hex(int(hb[w_])), "cpu", hex(cpu_hb(w_)))
    okm = np.nonzero((prev.astype(np.int64) == wp) & (wp >= 0))[0]
    print("   mismatches: want<1024:", int((wp[bad] < 1024).sum()), " same 1024-block:", int(((bad // 1024) == (wp[bad] // 1024)).sum()),
          "| matches with a prev: ", len(okm), "same 1024-block:", int(((okm // 1024) == (wp[okm] // 1024)).sum()),
          "| max idx of a match whose prev is in another block:", int(okm[(okm // 1024) != (wp[okm] // 1024)].max()) if ((okm // 1024) != (wp[okm] // 1024)).any() else None)
    if len(bad):
        print("   prev values on the device at the mismatches:", np.unique(prev[bad])[:5].tolist(), " distance idx-want: min", int((bad - wp[bad]).min()), "max", int((bad - wp[bad]).max()))
    wrong_dev = [i for i in range(0, min(n, 4096), 7) if int(hb[i]) != cpu_hb(i)]
    print("   records (every 7th of the first 4096) whose device hash bits differ from the CPU's:", len(wrong_dev), wrong_dev[:10])
    found = ctl[0]
    got = cuts[:found].tolist()
    same = got == want[:found]
    first_bad = next((k for k in range(min(found, len(want))) if got[k] != want[k]), None)
    print(dict(n=n, keys=keys, M=M, variant=variant, device=device, err=err, ctl=ctl[:5], found=found, want=len(want), same=same, first_bad=first_bad,
               got_head=got[:4], want_head=want[:4], around=(got[first_bad - 1:first_bad + 2], want[first_bad - 1:first_bad + 2]) if first_bad else None), flush=True)


case(600_000, 100_000, 5000, 5007, 1, True)
case(150_000, 3_000, 100, 107, 1, True)
case(2_000_000, 1_000_000, 5000, 2, 0, True)
