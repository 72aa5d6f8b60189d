#!/usr/bin/env python3
"""bench.py — flow-records/s of the MI355X flow-aggregation hot path.

One "step" = one pass of the hot path over one batch of synthetic input:
fold a Zipf(1.1) stream of 144-byte flow_record_t (already resident in HBM)
into the flow table (nfagg_ingest_device) and evict it (nfagg_evict_device).
Workload at N=1: BASELINE.json configs[1] — 100 M records, 1 M unique flows,
hash-aggregate only. At N>1 the records shard by flow-key hash: every rank owns
the population members whose key hashes to it and folds its own stream of the
same size (weak scaling, no data-path collective; --sketches adds the per-tick
RCCL all-reduce of the Count-Min / HLL arrays, configs[2]/[3]).

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit,
`roofline` (HBM-bound: algorithmic bytes / ingest-kernel time measured with HIP
events on the kernel's stream) and `cpu_baseline` (the CPU oracle — a C
restatement of pkg/flow.Accounter — timed on a bounded sample of the same stream).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_INGEST = 392      # 144 read record + 144 read slot + 104 write value
ALG_BYTES_SKETCH = 130      # CM 2 keys x 4 rows x (8+8) + HLL 2 x (1+1)
ALG_BYTES_EVICT = 296       # per evicted flow
DEFAULT_MAX_ENTRIES = 1 << 21   # CACHE_MAX_FLOWS of the bench table: SURVEY.md §8(d) config 2 sizing (2^22 slots = 1 GiB); tests/test_full_size_gpu.py uses the same
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=int, default=100_000_000, help="records per GPU per step")
    ap.add_argument("--flows", type=int, default=1_000_000, help="unique flows per GPU")
    ap.add_argument("--zipf", type=float, default=1.1)
    ap.add_argument("--hot-permille", type=int, default=0, help="configs[4]: share of records hitting one flow")
    ap.add_argument("--sketches", action="store_true", help="configs[2]/[3]: CM(d=4,w=2^20)+HLL(p=14), all-reduced per step when N>1")
    ap.add_argument("--variant", type=int, default=0, help="ingest kernel variant (DESIGN.md)")
    ap.add_argument("--dedup", action="store_true", help="configs[4]: NFAGG_MODE_KERNEL_DEDUP, every flow seen on two interfaces (stream variant 2)")
    ap.add_argument("--chunk", type=int, default=0, help="records per nfagg_ingest_device call (0 = whole stream)")
    ap.add_argument("--cpu-sample", type=int, default=20_000_000, help="records of the stream the CPU oracle is timed on (0 = skip)")
    ap.add_argument("--max-entries", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend at N>1: nccl (= RCCL, the product path) or gloo "
                    "(rehearsal of the N>1 code path on a 1-GPU box together with --same-device)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--group-local-fold", action="store_true", help="with --group-devices: NFAGG_GROUP_LOCAL_FOLD (no routing; every member "
                    "folds its own slice, the members' slots are merged into their owners at the eviction)")
    ap.add_argument("--group-devices", default="", help="ONE process driving several GPUs through nfagg_group_* (how the one-process Go agent "
                    "runs): comma-separated HIP ordinals, e.g. 0,1,2,3,4,5,6,7 — or 0,0,0,0 to rehearse four members on one GPU. One COMMON "
                    "stream (slice i arrives on member i's device), partitioned on the device and routed by key hash; not the torchrun contract path")
    args = ap.parse_args()
    if os.environ.get("NFAGG_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["NFAGG_BENCH_WATCHDOG"]), exit=True)

    import torch
    import torch.distributed as dist

    if args.group_devices:
        return group_main(args, torch)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    if rank == 0:
        import __graft_entry__
        __graft_entry__.ensure_built()      # fresh checkout: compile the HIP library first (git-ignored artefact)
    if world > 1:
        dist.barrier()
    import netobserv_ebpf_agent_amd as nf
    from netobserv_ebpf_agent_amd import synth

    n, keys = args.records, args.flows
    # ---- synthetic stream, generated in HBM (seed 2)
    th = synth.zipf_thresholds(keys, args.zipf)
    d_th = torch.from_numpy(th.view(np.int64)).cuda()
    d_pop = None
    if world > 1:
        pop = synth.shard_population(keys, world, rank)
        d_pop = torch.from_numpy(pop.view(np.int64)).cuda()
    d_recs = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    seed = 2 + 1000 * rank
    synth.stream_device(d_recs.data_ptr(), n, seed=seed, n_keys=keys, d_thresholds=d_th.data_ptr(),
                        hot_permille=args.hot_permille, variant=2 if args.dedup else 0,
                        d_pop_index=d_pop.data_ptr() if d_pop is not None else 0)
    torch.cuda.synchronize()

    # CACHE_MAX_FLOWS. The 100 M-record call has live + batch > max_entries, so the library folds it optimistically
    # (one fold, then the proof that no record found the table full: n_live <= max_entries) — DESIGN.md §2.
    max_entries = args.max_entries or DEFAULT_MAX_ENTRIES
    sk_flags = (nf.SKETCH_CM | nf.SKETCH_HLL) if args.sketches else 0
    ext = None
    cm_t = hll_t = None
    if args.sketches:
        cm_t = [torch.zeros(4 << 20, dtype=torch.int64, device="cuda") for _ in range(2)]
        hll_t = [torch.zeros(1 << 14, dtype=torch.int32, device="cuda") for _ in range(2)]
        ext = [cm_t[0].data_ptr(), cm_t[1].data_ptr(), hll_t[0].data_ptr(), hll_t[1].data_ptr()]
        torch.cuda.synchronize()
    tab = nf.FlowTable(max_entries=max_entries, device=local_rank, sketches=sk_flags, profile=True,
                       mode=nf.MODE_KERNEL_DEDUP if args.dedup else nf.MODE_ACCOUNTER,
                       ingest_variant=args.variant, n_shards=world, shard_id=rank, ext_sketch=ext)
    d_out = torch.empty(keys * 144 + 16, dtype=torch.uint8, device="cuda")
    chunk = args.chunk or n

    def step():
        off = 0
        while off < n:
            m = min(chunk, n - off)
            rc, c = tab.ingest_device(d_recs.data_ptr() + off * 144, m)
            assert rc == nf.OK and c == m, (rc, c)
            off += m
        if args.sketches and world > 1:
            tab.sync()          # the sketch kernels run on the table's stream
            nf.distributed.merge_sketches(cm_t, hll_t)
        flows = tab.evict_device(d_out.data_ptr(), keys, nf.REASON_TIMEOUT)
        if args.sketches:
            tab.sketch_reset()
        return flows

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    flows = 0
    for _ in range(args.warmup):
        flows = step()
    barrier()
    tab.reset_profile()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        flows = step()
    barrier()
    dt = time.perf_counter() - t0
    st = tab.stats()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        f = torch.tensor([flows], dtype=torch.int64, device="cuda")
        dist.all_reduce(f, op=dist.ReduceOp.SUM)
        flows_total = int(f.item())
    else:
        flows_total = flows

    if rank == 0:
        steps = max(args.steps, 1)
        total_records = n * world * steps
        ingest_ms = st.ingest_kernel_ms / max(st.ingest_launches, 1)
        recs_per_launch = n * steps / max(st.ingest_launches, 1)
        alg_bytes = ALG_BYTES_INGEST + (ALG_BYTES_SKETCH if args.sketches else 0)   # 392 B/record, 522 with the sketches
        achieved = alg_bytes * recs_per_launch / (ingest_ms * 1e-3) / 1e9 if ingest_ms > 0 else 0.0
        out = {
            "metric": "flow-records/s ingested + evictions/s, 1/2/4/8 GPU; % HBM roofline",
            "value": round(total_records / dt / 1e6, 3),
            "unit": "Mrecords/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": ("configs[%d]: %dM-record Zipf(%.1f) stream, %dk unique flows per GPU, hash-aggregate%s%s, device-resident input"
                             % (4 if args.dedup else (2 if args.sketches else 1), n // 1_000_000, args.zipf, keys // 1000,
                                "+CM(d=4,w=2^20)+HLL(p=14)" if args.sketches else " only",
                                ", kernel-dedup merge on" if args.dedup else "")),
                "records_per_gpu_per_step": n, "unique_flows_per_gpu": keys, "hot_permille": args.hot_permille,
                "stream_variant": 2 if args.dedup else 0, "mode": "kernel_dedup" if args.dedup else "accounter",
                "max_entries": max_entries, "table_bytes": int(st.table_bytes), "chunk": chunk,
                "parallelism": "key-hash shards x%d" % world + ("" if args.backend == "nccl" and not args.same_device
                                                                        else " (REHEARSAL: backend %s, same_device %s)" % (args.backend, args.same_device)), "ingest_variant": args.variant,
                "evictions_per_step": 1, "evicted_flows_per_step": flows_total,
                "evictions_per_s": round(steps / dt, 3), "evicted_flows_per_s": round(flows_total * steps / dt, 1),
                **({"skew_note": "every rank's stream has ITS OWN hot flow (per-rank populations): the load is balanced by construction. "
                                 "One node-wide hot flow under key-hash sharding lands on ONE GPU (SURVEY.md 8(e)); the one-process group "
                                 "spreads it with NFAGG_GROUP_LOCAL_FOLD (bench.py --group-devices ... --group-local-fold, DESIGN.md 7 a')"}
                   if world > 1 and args.hot_permille else {}),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_dedup_claim + k_dedup_fold (one ingest call)" if args.dedup else
                           "part::k_pass1 + part::k_pass2 (+ k_merge_overflow + k_finalize) = one hash-insert/fold call" if args.variant == 0 else
                           "ingest variant %d" % args.variant),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                # honest yardsticks next to SURVEY §8(d)'s algorithmic figure (which charges a slot read + write per record that
                # the LDS flow cache never performs): frac_stream_floor = the 144-byte records alone, read once, against the
                # peak; frac_traffic = HBM bytes the counters saw (roofline.traffic), against the peak
                "frac_stream_floor": round(144 * recs_per_launch / (ingest_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ingest_ms > 0 else None,
                "frac_traffic": None,
                "alg_bytes_per_record": alg_bytes, "records_per_launch": int(recs_per_launch),
                "launch_ms": round(ingest_ms, 4), "launches": int(st.ingest_launches),
                "lds_cache_hit_rate": round(1.0 - st.records_bypassed / max(1, n * (args.steps + args.warmup)), 4),
                "kernel_Mrecords_per_s": round(recs_per_launch / (ingest_ms * 1e-3) / 1e6, 1) if ingest_ms > 0 else None,
                "evict_launch_ms": round(st.evict_kernel_ms / max(st.evict_launches, 1), 4),
                "sketch_launch_ms": round(st.sketch_kernel_ms / max(st.sketch_launches, 1), 4) if st.sketch_launches else None,
            },
        }
        ev_ms = st.evict_kernel_ms / max(st.evict_launches, 1)
        if ev_ms > 0 and not args.dedup:
            ev_ach = ALG_BYTES_EVICT * (flows_total / world) / (ev_ms * 1e-3) / 1e9
            out["roofline_evict"] = {"bound": "hbm", "kernel": "k_evict", "achieved": round(ev_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(ev_ach / HBM_PEAK_GBS, 4), "alg_bytes_per_flow": ALG_BYTES_EVICT,
                                     "flows_per_launch": int(flows_total / world), "launch_ms": round(ev_ms, 4), "traffic": None}
        # ---- HBM traffic of one ingest call, from rocprofv3 PMC passes of this same command line
        # (tools/profile_bench.sh + tools/summarize_prof.py -> profiles/<tag>_traffic.json; FETCH_SIZE/WRITE_SIZE
        # calibrated on known byte counts in the same session). Reported only for the workload it was measured on.
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
            try:
                tj = json.load(open(tf))
            except Exception:
                continue
            same = all(tj.get(k) == out["config"].get(k) for k in ("workload", "hot_permille", "stream_variant", "mode", "max_entries"))
            if same and tj.get("records_per_call") == int(recs_per_launch) and args.variant == 0 and ingest_ms > 0:
                out["roofline"]["traffic"] = round(tj["traffic_bytes_per_call"] / (ingest_ms * 1e-3) / 1e9, 1)
                out["roofline"]["frac_traffic"] = round(out["roofline"]["traffic"] / HBM_PEAK_GBS, 4)
                out["roofline"]["traffic_bytes_per_launch"] = int(tj["traffic_bytes_per_call"])
                out["roofline"]["traffic_source"] = os.path.relpath(tf, ROOT)
                if "roofline_evict" in out and tj.get("evict_traffic_bytes_per_call") and tj.get("evicted_flows") == out["roofline_evict"]["flows_per_launch"]:
                    out["roofline_evict"]["traffic"] = round(tj["evict_traffic_bytes_per_call"] / (ev_ms * 1e-3) / 1e9, 1)
                    out["roofline_evict"]["traffic_bytes_per_flow"] = round(tj["evict_traffic_bytes_per_call"] / tj["evicted_flows"], 1)
                break
        # ---- CPU baseline: the oracle (C restatement of pkg/flow.Accounter), 1 core, bounded sample
        if args.cpu_sample > 0 and world == 1:     # rank 0 at N=1 only
            from oracle import oracle as O
            O.build()
            m = min(args.cpu_sample, n)
            sample = d_recs[: m * 144].cpu().numpy()
            acc = O.Accounter(max_entries, 1 if args.dedup else 0)
            t1 = time.perf_counter()
            consumed = acc.ingest(sample)
            ev = acc.evict()
            cpu_dt = time.perf_counter() - t1
            acc.close()
            assert consumed == m
            out["cpu_baseline"] = {
                "value": round(m / cpu_dt / 1e6, 3), "unit": "Mrecords/s", "cores": 1, "kind": "port", "what": "C restatement of pkg/flow.Accounter (oracle/nfagg_oracle.c); the Go reference cannot be built here",
                "sample": "first %d records of rank 0's stream (%d flows), oracle Accounter ingest+evict, %.1f s" % (m, len(ev), cpu_dt),
                "host_cores_available": os.cpu_count(),
            }
            # best-effort multi-core variant of the same restatement: the sample split by a key hash
            # over T workers, one oracle Accounter each (ctypes releases the GIL). The reference itself is one goroutine.
            if not args.dedup:
                import threading
                T = max(2, min(32, (os.cpu_count() or 2) // 2))
                accs = [O.Accounter(max_entries, 0) for _ in range(T)]
                got = [0] * T

                def work(k):
                    got[k] = accs[k].ingest_shard(sample, T, k)
                    got[k] = (got[k], len(accs[k].evict()))
                t1 = time.perf_counter()
                ths = [threading.Thread(target=work, args=(k,)) for k in range(T)]
                for t_ in ths:
                    t_.start()
                for t_ in ths:
                    t_.join()
                mc_dt = time.perf_counter() - t1
                for a_ in accs:
                    a_.close()
                assert sum(g[0] for g in got) == m and sum(g[1] for g in got) == len(ev)
                out["cpu_baseline"]["multicore"] = {"value": round(m / mc_dt / 1e6, 3), "unit": "Mrecords/s", "cores": T, "kind": "port",
                                                    "sample": "same sample, key-hash split over %d threads, %.1f s" % (T, mc_dt)}
        # ---- the chip's copy bandwidth in this very run (state the peak used, confirm it with a D2D copy):
        # up to 4 GiB device-to-device (the stream's first half over its second: nothing reads the stream after this point),
        # read + written bytes over the time of the copy
        try:
            nb = min(4 << 30, (n * 144) // 2 // 256 * 256)
            src_t, dst_t = d_recs[:nb], d_recs[nb:2 * nb]
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dst_t.copy_(src_t); torch.cuda.synchronize()
            ev0.record(); dst_t.copy_(src_t); ev1.record(); torch.cuda.synchronize()
            out["roofline"]["hbm_copy_measured_GBs"] = round(2 * nb / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1)
        except Exception as exc:                      # a diagnostic, never a reason to lose the bench line
            out["roofline"]["hbm_copy_measured_GBs"] = None
            out["roofline"]["hbm_copy_error"] = str(exc)[:100]
        print(json.dumps(out), flush=True)
    tab.close()
    if world > 1:
        dist.destroy_process_group()


def group_main(args, torch):
    """--group-devices: one process, D members, ONE common stream of D x --records records over D x --flows flows. Slice i of the
    stream (arrival order: slice 0 first) is resident on member i's device — as if it had come up that member's PCIe link — and
    enters through nfagg_group_ingest_device(i, ...): stable device partition by key-hash shard, buckets to their owners
    (hipMemcpyPeerAsync over xGMI between distinct devices), per-member folds; per step one sketch merge (RCCL all-reduce when
    the devices are distinct) and one eviction of every shard."""
    import __graft_entry__
    __graft_entry__.ensure_built()
    import netobserv_ebpf_agent_amd as nf
    from netobserv_ebpf_agent_amd import synth
    devs = [int(x) for x in args.group_devices.split(",")]
    D, n, keys = len(devs), args.records, args.flows * len(devs)
    th = synth.zipf_thresholds(keys, args.zipf)
    slices, outs = [], []
    for i, d in enumerate(devs):
        torch.cuda.set_device(d)
        d_th = torch.from_numpy(th.view(np.int64)).cuda()
        buf = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        synth.stream_device(buf.data_ptr(), n, j0=i * n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr(), hot_permille=args.hot_permille, variant=0)
        torch.cuda.synchronize()
        slices.append(buf)
        outs.append(torch.empty((2 * args.flows + 4096) * 144, dtype=torch.uint8, device="cuda"))
    sk_flags = (nf.SKETCH_CM | nf.SKETCH_HLL) if args.sketches else 0
    max_entries = (args.max_entries or DEFAULT_MAX_ENTRIES) * D
    if args.group_local_fold:
        # max_entries stays D x the single-GPU table: it bounds every member undivided, and each member may see every flow
        outs = [torch.empty((2 * keys // D + 4096) * 144, dtype=torch.uint8, device="cuda:%d" % d) for d in devs]
    grp = nf.FlowGroup(devs, max_entries=max_entries, sketches=sk_flags, profile=True, local_fold=args.group_local_fold)
    out_cap = [o.numel() // 144 for o in outs]

    def step():
        for i in range(D):
            rc, c = grp.ingest_device(i, slices[i].data_ptr(), n)
            assert rc == nf.OK and c == n, (rc, c)
        if args.sketches:
            grp.merge_sketches()
        got = grp.evict_device([o.data_ptr() for o in outs], out_cap, nf.REASON_TIMEOUT)
        if args.sketches:
            for m in grp.members:
                m.sketch_reset()
        return sum(got)

    def sync_all():
        for d in set(devs):
            torch.cuda.synchronize(d)

    flows = 0
    for _ in range(args.warmup):
        flows = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        flows = step()
    sync_all()
    dt = time.perf_counter() - t0
    steps = max(args.steps, 1)
    sts = [m.stats() for m in grp.members]
    fold_ms = [st.ingest_kernel_ms / max(st.ingest_launches, 1) for st in sts]
    out = {
        "metric": "flow-records/s ingested + evictions/s, 1/2/4/8 GPU; % HBM roofline",
        "value": round(n * D * steps / dt / 1e6, 3), "unit": "Mrecords/s", "n_gpus": len(set(devs)), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "configs[3] shape through nfagg_group_*: ONE common %dM-record Zipf(%.1f) stream over %dk flows, %d members on devices %s, "
                        "%s%s, device-resident input" % (n * D // 1_000_000, args.zipf, keys // 1000, D, devs,
                                                         "local fold, raw slots merged into their owners at the eviction" if args.group_local_fold
                                                         else "device partition + routing by key hash",
                                                         ", CM+HLL merged per step" if args.sketches else ""),
            "group_mode": "local_fold" if args.group_local_fold else "routed",
            "members": D, "devices": devs, "records_per_member_slice": n, "unique_flows_total": keys, "max_entries_total": max_entries,
            "evicted_flows_per_step": flows, "parallelism": "one process, group of %d members (distinct devices: %s)" % (D, len(set(devs)) == D),
            "member_fold_ms_per_launch": [round(x, 3) for x in fold_ms],
            "member_records_ingested": [int(st.records_ingested) for st in sts],
        },
    }
    print(json.dumps(out), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
