#!/usr/bin/env python3
"""bench.py — flow-records/s of the MI355X flow-aggregation hot path.

One "step" = one pass of the hot path over one batch of synthetic input: fold a Zipf(1.1) stream of 144-byte
flow_record_t (already resident in HBM) into the flow table and evict it.

N = 1 (default): BASELINE.json configs[1] — 100 M records, 1 M unique flows, hash-aggregate only: ONE
  nfagg_ingest_device call + one nfagg_evict_device per step. After the timed steps the same run measures, bounded to a few
  seconds, the other legs the reference's users care about (`extra`): the PCIe-inclusive host path, configs[2] (+ Count-Min
  + HLL), the configs[4] shape (dedup merge on, 90 % of the records one flow) and the reference's default CACHE_MAX_FLOWS.

N > 1: BASELINE.json configs[3] — ONE common stream of N x 125 M records over N x 1.25 M flows, Count-Min + HLL on, one
  process per GPU (torch.distributed, backend nccl = RCCL). Slice r of the stream (arrival positions [r n, (r+1) n)) is
  resident on rank r, as if it had come up that GPU's PCIe link. Nothing is pre-sharded: every rank folds what arrived at
  it, whatever its keys, with sequence numbers global to the job (LOCAL FOLD), and per step — inside the
  timed region — the sketches are all-reduced (RCCL, sum u64 / max u32), every rank's flows travel as 192-byte partials to
  the rank that owns them (nfagg_shard_of; RCCL all-to-all over xGMI), the owners merge and evict. The union of the ranks'
  evictions is bit-identical to one Accounter over the whole stream (tests/test_partials_gpu.py).
  `python bench.py --gpus N` as a plain process spawns its N ranks itself (python -m torch.distributed.run); launched by
  torch.distributed.run it is one of them. --presharded keeps round 2's communication-free line (every rank folds a
  private stream over its own shard's population: linear by construction) for comparison.

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit, `roofline` (HBM-bound: algorithmic bytes /
ingest-kernel time measured with HIP events on the kernel's stream) and, at N = 1, `cpu_baseline` (the CPU oracle — a C
restatement of pkg/flow.Accounter — timed on a bounded sample of the same stream).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_INGEST = 392      # 144 read record + 144 read slot + 104 write value
ALG_BYTES_SKETCH = 130      # CM 2 keys x 4 rows x (8+8) + HLL 2 x (1+1)
ALG_BYTES_EVICT = 296       # per evicted flow
DEFAULT_MAX_ENTRIES = 1 << 21   # CACHE_MAX_FLOWS of the bench table: SURVEY.md §8(d) config 2 sizing (2^22 slots = 768 MiB); tests/test_full_size_gpu.py uses the same
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
METRIC = "flow-records/s ingested + evictions/s, 1/2/4/8 GPU; % HBM roofline"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=int, default=0, help="records per GPU per step (0: 100 M at N = 1, 125 M at N > 1)")
    ap.add_argument("--flows", type=int, default=0, help="unique flows per GPU (0: 1 M at N = 1, 1.25 M at N > 1)")
    ap.add_argument("--zipf", type=float, default=1.1)
    ap.add_argument("--hot-permille", type=int, default=0, help="configs[4]: share of records hitting one flow")
    ap.add_argument("--sketches", action="store_true", help="configs[2]: CM(d=4,w=2^20)+HLL(p=14) at N = 1 (always on at N > 1 unless --no-sketches)")
    ap.add_argument("--no-sketches", action="store_true", help="N > 1 without the sketches and their all-reduce")
    ap.add_argument("--variant", type=int, default=0, help="ingest kernel variant (DESIGN.md)")
    ap.add_argument("--dedup", action="store_true", help="configs[4]: NFAGG_MODE_KERNEL_DEDUP, every flow seen on two interfaces (stream variant 2)")
    ap.add_argument("--chunk", type=int, default=0, help="records per nfagg_ingest_device call (0 = whole stream)")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="records of the stream the CPU oracle is timed on (0 = skip; default: the whole "
                    "100 M-record stream of configs[1] — ~10 s on one core, ~4 s more for the multi-core variants)")
    ap.add_argument("--max-entries", type=int, default=0)
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the bounded extra legs (e2e host path, configs[2], configs[4] shape, CACHE_MAX_FLOWS 5000)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1 (local fold): window k's tick (sketch all-reduce, partials export, all-to-all, "
                    "merge, eviction) strictly after its fold and before the next window's — round 5's line. Default: the tick of window k runs "
                    "beside the fold of window k + 1 (two tables per rank, a host thread for the tick)")
    ap.add_argument("--dump-evictions", default="", help="tests: after the timed windows every rank writes the records of its LAST eviction to "
                    "<path>.<rank> (raw 144-byte records): their union is compared with ONE oracle Accounter over the common stream")
    ap.add_argument("--presharded", action="store_true", help="N > 1: round 2's line — every rank folds a private stream over its own shard's "
                    "population through the shard filter; no data-path exchange (linear by construction; for comparison only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend at N>1: nccl (= RCCL, the product path) or gloo "
                    "(rehearsal of the N>1 code path on a 1-GPU box together with --same-device)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true", help="N = 1 through the N > 1 code path: init_process_group(backend) with ONE rank, the sketch "
                    "all-reduce (int64 SUM, uint8 MAX), all_to_all_single with split sizes on device tensors, partials export / merge / "
                    "evict_owned — the multi-GPU preflight on the one GPU there is (tests/test_dist_preflight_gpu.py)")
    ap.add_argument("--group-local-fold", action="store_true", help="with --group-devices: NFAGG_GROUP_LOCAL_FOLD (no routing; every member "
                    "folds its own slice, the members' slots are merged into their owners at the eviction)")
    ap.add_argument("--group-threads", action="store_true", help="with --group-devices --group-local-fold: one host thread per member "
                    "(the members' folds overlap; the Go host would use one goroutine per GPU)")
    ap.add_argument("--group-devices", default="", help="ONE process driving several GPUs through nfagg_group_* (how the one-process Go agent "
                    "runs): comma-separated HIP ordinals, e.g. 0,1,2,3,4,5,6,7 — or 0,0,0,0 to rehearse four members on one GPU. One COMMON "
                    "stream (slice i arrives on member i's device), partitioned on the device and routed by key hash; not the torchrun contract path")
    ap.add_argument("--print-spawn", action="store_true", help="print the command line --gpus N would spawn and exit (no GPU needed)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_command(argv, gpus, port):
    """`python bench.py --gpus N` as a plain process: the ranks are spawned exactly as the driver's contract launches them."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--print-spawn"]


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if os.environ.get("NFAGG_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["NFAGG_BENCH_WATCHDOG"]), exit=True)
    if args.print_spawn:
        print(" ".join(spawn_command(argv, args.gpus, 29500)))
        return 0
    if args.group_devices:
        import torch
        return group_main(args, torch)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain process: become the launcher of N ranks (one per GPU) — the same command line the driver's contract uses
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        import __graft_entry__
        __graft_entry__.ensure_built()            # once, before N ranks race for it
        return subprocess.call(spawn_command(argv, args.gpus, free_port()), env=env)
    return rank_main(args)


def frac_of_a_real_bound(rf):
    """SURVEY 8(d)'s algorithmic bytes over the launch time are a MODEL rate (a slot read + write per record that a fold in LDS never
    moves). Where it exceeds the peak it describes no bound: achieved / frac become those of a real one — the counter traffic when a
    PMC file of this workload taken on the loaded library exists, else the records read once — and frac_basis says which; the model
    stays under alg_model_GBs (round-5 review, item 3: no frac above 1 anywhere in a bench line)."""
    if rf.get("frac") is None or rf["frac"] <= 1.0:
        return rf
    if rf.get("traffic") and not rf.get("traffic_stale"):
        rf["achieved"], rf["frac"], rf["frac_basis"] = rf["traffic"], rf["frac_traffic"], "counter_traffic"
    elif rf.get("frac_stream_floor") is not None:
        rf["achieved"], rf["frac"], rf["frac_basis"] = round(rf["frac_stream_floor"] * HBM_PEAK_GBS, 1), rf["frac_stream_floor"], "stream_floor"
    return rf


class _QuietStdout:
    """stdout carries ONE JSON line (the driver's contract). Whatever libraries print there while the bench runs — RCCL's
    version banner at communicator creation, a compiler invoked by ensure_built() — is sent to stderr instead: file
    descriptor 1 points at stderr until emit() restores it for the line."""

    def __init__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        print(line, flush=True)
        os.dup2(2, 1)


def resolve_sizes(args, world):
    n = args.records or (100_000_000 if world == 1 else 125_000_000)
    keys = args.flows or (1_000_000 if world == 1 else 1_250_000)
    return n, keys


def next_pow2(v):
    p = 1
    while p < v:
        p <<= 1
    return p


def rank_main(args):
    import torch
    import torch.distributed as dist

    quiet = _QuietStdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d" % (args.gpus, world))
    dist_on = world > 1 or args.force_dist          # --force-dist: ONE rank through everything the N > 1 path does
    if args.force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    if args.same_device:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (rehearse the N > 1 path on one GPU with --same-device --backend gloo)"
                         % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    if rank == 0:
        import __graft_entry__
        __graft_entry__.ensure_built()      # fresh checkout: compile the HIP library first (git-ignored artefact)
    if dist_on:
        dist.barrier()
    import netobserv_ebpf_agent_amd as nf
    from netobserv_ebpf_agent_amd import synth

    n, keys = resolve_sizes(args, world)
    # N > 1: ONE common stream, local fold — in both modes. With --dedup (configs[4]) the ranks' tables are keyed by (flow,
    # interface) and the flows are put together at their owners when the epoch ends (nfagg_config.local_fold, DESIGN.md §7 a')
    local_fold = dist_on and not args.presharded
    sketches = args.sketches or (dist_on and not args.no_sketches)
    keys_total = keys * world if local_fold else keys
    # ---- synthetic stream, generated in HBM (seed 2)
    th = synth.zipf_thresholds(keys_total, args.zipf)
    d_th = torch.from_numpy(th.view(np.int64)).cuda()
    d_pop = None
    j0, seed = 0, 2
    if local_fold:
        j0 = rank * n                           # ONE stream of world x n records: this rank holds arrival positions [rank n, (rank + 1) n)
    elif dist_on:
        pop = synth.shard_population(keys, world, rank)
        d_pop = torch.from_numpy(pop.view(np.int64)).cuda()
        seed = 2 + 1000 * rank
    d_recs = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()

    def gen_stream(variant, hot):
        synth.stream_device(d_recs.data_ptr(), n, j0=j0, seed=seed, n_keys=keys_total, d_thresholds=d_th.data_ptr(),
                            hot_permille=hot, variant=variant, d_pop_index=d_pop.data_ptr() if d_pop is not None else 0)
        torch.cuda.synchronize()

    gen_stream(2 if args.dedup else 0, args.hot_permille)

    # CACHE_MAX_FLOWS. The 100 M-record call has live + batch > max_entries, so the library folds it optimistically
    # (one fold, then the proof that no record found the table full: n_live <= max_entries) — DESIGN.md §2.
    # Local fold: a rank may see any flow of the stream (5.4 M of 10 M at N = 8), and its table also takes the flows it owns
    # from the other ranks.
    # (kernel-dedup local fold: the table holds one slot per (flow, interface), two interfaces per flow in stream variant 2)
    max_entries = args.max_entries or (DEFAULT_MAX_ENTRIES if not local_fold else
                                       max(DEFAULT_MAX_ENTRIES, min(next_pow2(keys_total * (2 if args.dedup else 1)), 1 << (24 if args.dedup else 23))))
    sk_flags = (nf.SKETCH_CM | nf.SKETCH_HLL) if sketches else 0
    # Local fold at N > 1: TWO tables per rank, windows alternate between them — while window k + 1 is folded into one, window k's tick
    # (sketch all-reduce, partials to their owners, merge, eviction) runs on the other, from a host thread of its own (SURVEY.md
    # §8(e); --no-overlap: one table, everything in sequence). Each table has its own sketch arrays: the all-reduce
    # of window k's must not see window k + 1's records.
    overlap = local_fold and not args.no_overlap
    n_tabs = 2 if overlap else 1
    tabs, cm_ts, hll_ts = [], [], []
    for _ in range(n_tabs):
        ext = None
        cm_t = hll_t = None
        if sketches:
            cm_t = [torch.zeros(4 << 20, dtype=torch.int64, device="cuda") for _ in range(2)]
            hll_t = [torch.zeros(1 << 14, dtype=torch.uint8, device="cuda") for _ in range(2)]     # one byte per register: a 16 KiB all-reduce
            ext = [cm_t[0].data_ptr(), cm_t[1].data_ptr(), hll_t[0].data_ptr(), hll_t[1].data_ptr()]
            torch.cuda.synchronize()
        tabs.append(nf.FlowTable(max_entries=max_entries, device=local_rank, sketches=sk_flags, profile=True,
                                 mode=nf.MODE_KERNEL_DEDUP if args.dedup else nf.MODE_ACCOUNTER,
                                 ingest_variant=args.variant, n_shards=1 if local_fold else world, shard_id=0 if local_fold else rank, ext_sketch=ext,
                                 local_fold=local_fold))
        cm_ts.append(cm_t); hll_ts.append(hll_t)
    tab = tabs[0]
    PARTIAL_BYTES = tab.partial_bytes           # 192; 256 for the sub-flow partials of the kernel-dedup mode
    out_cap = (keys_total if local_fold else keys) + 4096
    d_out = torch.empty(min(out_cap, max_entries + 4096) * 144 + 16, dtype=torch.uint8, device="cuda")
    out_cap = (d_out.numel() - 16) // 144
    chunk = args.chunk or n
    # local fold: partials out (at most max_entries flows live here) and in (what this rank owns of the others' flows)
    d_exp = d_imp = None
    if local_fold:
        d_exp = torch.empty(max_entries * (PARTIAL_BYTES // 8), dtype=torch.int64, device="cuda")
        d_imp = torch.empty(max_entries * (PARTIAL_BYTES // 8), dtype=torch.int64, device="cuda")
    phase = {}

    def exchange(counts):
        """Segment o of this rank's export goes to rank o: RCCL all-to-all over xGMI (through host memory in the gloo rehearsal)."""
        W8 = PARTIAL_BYTES // 8
        send_counts = torch.tensor(counts, dtype=torch.int64)
        recv_counts = torch.empty(world, dtype=torch.int64)
        if args.backend == "nccl":
            sc, rc_ = send_counts.cuda(), recv_counts.cuda()
            dist.all_to_all_single(rc_, sc)
            recv_counts = rc_.cpu()
        else:
            dist.all_to_all_single(recv_counts, send_counts)
        ins = [int(c) * W8 for c in counts]
        outs = [int(c) * W8 for c in recv_counts.tolist()]
        total_in = sum(outs)
        assert total_in <= d_imp.numel(), "partials landing area too small"
        send, recv = d_exp[: sum(ins)], d_imp[:total_in]
        if args.backend == "nccl":
            dist.all_to_all_single(recv, send, outs, ins)
        else:
            r_cpu = torch.empty(total_in, dtype=torch.int64)
            dist.all_to_all_single(r_cpu, send.cpu(), outs, ins)
            recv.copy_(r_cpu)
        torch.cuda.synchronize()
        return total_in // W8, sum(counts)

    def fold(w):
        tb = tabs[w % n_tabs]
        if local_fold:
            tb.set_sequence(rank * n)
        off = 0
        while off < n:
            m = min(chunk, n - off)
            rc, c = tb.ingest_device(d_recs.data_ptr() + off * 144, m)
            assert rc == nf.OK and c == m, (rc, c)
            off += m

    def tick(w, timed=None):
        tb, cm_t, hll_t = tabs[w % n_tabs], cm_ts[w % n_tabs], hll_ts[w % n_tabs]

        def mark(name, t_prev):
            if timed is None:
                return t_prev
            torch.cuda.synchronize(); tb.sync()
            t = time.perf_counter()
            timed[name] = timed.get(name, 0.0) + (t - t_prev) * 1e3
            return t
        t = time.perf_counter()
        if sketches and dist_on:
            tb.sync()          # the sketch kernels run on the table's stream
            nf.distributed.merge_sketches(cm_t, hll_t)
            torch.cuda.synchronize()
        t = mark("sketch_allreduce_ms", t)
        if local_fold:
            rc, counts, n_exp = tb.partials_export_device(world, rank, d_exp.data_ptr(), d_exp.numel() * 8 // PARTIAL_BYTES)
            assert rc == nf.OK, "partials buffer too small: %d needed" % n_exp
            t = mark("export_ms", t)
            n_in, n_sent = exchange(counts)
            t = mark("all_to_all_ms", t)
            tb.partials_merge_device(world, rank, d_imp.data_ptr(), n_in)
            rc, flows = tb.evict_owned_device(world, rank, d_out.data_ptr(), out_cap, nf.REASON_TIMEOUT)
            assert rc == nf.OK, "eviction buffer too small: %d needed" % flows
            t = mark("merge_evict_ms", t)
            if timed is not None:
                timed["partials_sent"], timed["partials_received"] = n_sent, n_in
        else:
            flows = tb.evict_device(d_out.data_ptr(), out_cap, nf.REASON_TIMEOUT)
            t = mark("evict_ms", t)
        if sketches:
            tb.sketch_reset()
        return flows

    def step(timed=None):
        """One window, everything in sequence (N = 1; --no-overlap; the instrumented step)."""
        t = time.perf_counter()
        fold(0)
        if timed is not None:
            torch.cuda.synchronize(); tabs[0].sync()
            timed["fold_ms"] = timed.get("fold_ms", 0.0) + (time.perf_counter() - t) * 1e3
        return tick(0, timed)

    def run_windows(k_windows, timed=None):
        """k_windows windows. Overlapped: the tick of window w runs on a host thread beside the fold of window w + 1 (other table,
        other sketch arrays; the library's streams are non-blocking and the collectives run on torch's: nothing serialises them);
        the last tick has no fold beside it. One tick at a time: the next starts when the previous has ended."""
        if not overlap:
            f = 0
            for _ in range(k_windows):
                f = step()
            return f
        import threading
        res = {"flows": 0, "err": None}

        def tick_thread(w):
            try:
                torch.cuda.set_device(local_rank)
                t0_ = time.perf_counter()
                res["flows"] = tick(w)
                if timed is not None:
                    timed["tick_ms_beside_a_fold"] = timed.get("tick_ms_beside_a_fold", 0.0) + (time.perf_counter() - t0_) * 1e3
            except BaseException as exc:
                res["err"] = exc
        th = None
        for w in range(k_windows):
            t0_ = time.perf_counter()
            fold(w)
            tabs[w % n_tabs].sync()
            if timed is not None:
                timed["fold_ms_beside_a_tick"] = timed.get("fold_ms_beside_a_tick", 0.0) + (time.perf_counter() - t0_) * 1e3
            if th is not None:
                th.join()
                if res["err"] is not None:
                    raise res["err"]
            th = threading.Thread(target=tick_thread, args=(w,))
            th.start()
        th.join()
        if res["err"] is not None:
            raise res["err"]
        return res["flows"]

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        for tb in tabs:
            tb.sync()

    flows = 0
    if args.warmup:
        flows = run_windows(args.warmup)
    barrier()
    for tb in tabs:
        tb.reset_profile()
    t0 = time.perf_counter()
    flows = run_windows(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if args.dump_evictions:
        d_out[: flows * 144].cpu().numpy().tofile("%s.%d" % (args.dump_evictions, rank))
    class _Sum:                                     # the tables' stats, added up (two tables when the windows alternate)
        pass
    st = _Sum()
    sts = [tb.stats() for tb in tabs]
    for name in ("records_ingested", "ingest_kernel_ms", "ingest_launches", "sketch_kernel_ms", "sketch_launches", "evict_kernel_ms",
                 "evict_launches", "records_bypassed", "table_bytes"):
        setattr(st, name, sum(getattr(x, name) for x in sts))
    records_folded = [int(st.records_ingested)]
    if dist_on:
        mdev = "cuda" if args.backend == "nccl" else "cpu"          # bookkeeping collectives (gloo rehearsal: host tensors)
        t = torch.tensor([dt], dtype=torch.float64, device=mdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        f = torch.tensor([flows], dtype=torch.int64, device=mdev)
        dist.all_reduce(f, op=dist.ReduceOp.SUM)
        flows_total = int(f.item())
        rf = [torch.zeros(1, dtype=torch.int64, device=mdev) for _ in range(world)]
        dist.all_gather(rf, torch.tensor([int(st.records_ingested)], dtype=torch.int64, device=mdev))
        records_folded = [int(x.item()) for x in rf]
        # one more, instrumented window (outside the timed region), everything in sequence: what each phase costs on its own ...
        step(timed=phase)
        barrier()
        if overlap:
            # ... and three more, overlapped as the timed ones: how long a fold and a tick take when they run beside each other
            ov = {}
            t_ov = time.perf_counter()
            run_windows(3, timed=ov)
            barrier()
            phase["overlapped"] = {"windows": 3, "wall_ms_per_window": round((time.perf_counter() - t_ov) * 1e3 / 3, 3),
                                   "fold_ms_beside_a_tick": round(ov.get("fold_ms_beside_a_tick", 0.0) / 3, 3),
                                   "tick_ms_beside_a_fold": round(ov.get("tick_ms_beside_a_fold", 0.0) / 3, 3),
                                   "what": "window w's tick (all-reduce, export, all-to-all, merge, evict) on a host thread beside window w + 1's fold; "
                                           "the exchange is hidden when wall_ms_per_window is about max(fold, tick), not their sum"}
    else:
        flows_total = flows

    if rank == 0:
        steps = max(args.steps, 1)
        total_records = n * world * steps
        ingest_ms = st.ingest_kernel_ms / max(st.ingest_launches, 1)
        # a sketch update that runs as a launch of its own (small kernel-dedup batches; the cached dedup fold and the accounter
        # fold feed the sketches themselves) belongs to the call's time
        sketch_sep_ms = st.sketch_kernel_ms / max(st.ingest_launches, 1) if st.sketch_launches else 0.0
        ingest_ms += sketch_sep_ms
        recs_per_launch = n * steps / max(st.ingest_launches, 1)
        alg_bytes = ALG_BYTES_INGEST + (ALG_BYTES_SKETCH if sketches else 0)   # 392 B/record, 522 with the sketches
        achieved = alg_bytes * recs_per_launch / (ingest_ms * 1e-3) / 1e9 if ingest_ms > 0 else 0.0
        cfg_no = 4 if args.dedup else (3 if dist_on else (2 if sketches else 1))
        if local_fold and args.dedup:
            workload = ("configs[4]: ONE %dM-record stream, %d permille of the records one flow alternating over two interfaces, the rest "
                        "Zipf(%.1f) over %dk unique flows on two interfaces each, %dM records per GPU resident on the GPU they arrived at, "
                        "kernel-dedup merge on (bpf/flows.c:76-143), local fold over sub-flow tables (no per-record routing: the hot flow "
                        "is folded by every GPU)%s + per step: sub-flow partials to the owners of their flows (all-to-all), merge, join, eviction"
                        % (n * world // 1_000_000, args.hot_permille, args.zipf, keys_total // 1000, n // 1_000_000,
                           ", RCCL all-reduce of CM+HLL" if sketches else ""))
        elif local_fold:
            workload = ("configs[3]: ONE %dM-record Zipf(%.1f) stream over %dk unique flows, %dM records per GPU resident on the GPU they "
                        "arrived at, local fold (no per-record routing) + per step: RCCL all-reduce of CM(d=4,w=2^20)+HLL(p=14), flow "
                        "partials to their key-hash owners (all-to-all), merge, eviction"
                        % (n * world // 1_000_000, args.zipf, keys_total // 1000, n // 1_000_000))
        else:
            workload = ("configs[%d]: %dM-record Zipf(%.1f) stream, %dk unique flows per GPU, hash-aggregate%s%s, device-resident input"
                        % (cfg_no, n // 1_000_000, args.zipf, keys // 1000, "+CM(d=4,w=2^20)+HLL(p=14)" if sketches else " only",
                           ", kernel-dedup merge on" if args.dedup else ""))
        rehearsal = "" if (args.backend == "nccl" and not args.same_device) else " (REHEARSAL: backend %s, same_device %s)" % (args.backend, args.same_device)
        out = {
            "metric": METRIC,
            "value": round(total_records / dt / 1e6, 3),
            "unit": "Mrecords/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": workload,
                "records_per_gpu_per_step": n, "unique_flows_per_gpu": keys, "hot_permille": args.hot_permille,
                "stream_variant": 2 if args.dedup else 0, "mode": "kernel_dedup" if args.dedup else "accounter",
                "max_entries": max_entries, "table_bytes": int(st.table_bytes), "chunk": chunk,
                "parallelism": ("one process per GPU x%d, local fold + partials to key-hash owners%s" % (world, ", window w's tick beside window "
                                 "w + 1's fold (two tables per rank)" if overlap else ", fold and tick in sequence") if local_fold
                                else "key-hash shards x%d" % world) + rehearsal,
                "windows_overlapped": bool(overlap),
                "ingest_variant": args.variant,
                "evictions_per_step": 1, "evicted_flows_per_step": flows_total,
                "evictions_per_s": round(steps / dt, 3), "evicted_flows_per_s": round(flows_total * steps / dt, 1),
                **({"skew_note": "every rank's stream has ITS OWN hot flow (per-rank populations): the load is balanced by construction. "
                                 "One node-wide hot flow under key-hash sharding lands on ONE GPU (SURVEY.md 8(e)); the local fold "
                                 "spreads it (default at N > 1)"}
                   if world > 1 and args.hot_permille and args.presharded else {}),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_dedup_stream + k_dedup_parts (+ overflow kernels + k_finalize) = one kernel-dedup ingest call" if args.dedup else
                           "part::k_pass1 + part::k_pass2 (+ k_merge_overflow + k_finalize) = one hash-insert/fold call" if args.variant == 0 else
                           "ingest variant %d" % args.variant),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "frac_basis": "alg_model",
                "alg_model_GBs": round(achieved, 1),
                "frac_note": "achieved / peak as the bench contract defines it: SURVEY 8(d)'s algorithmic bytes (a slot read + write per record, which "
                             "a fold in LDS never moves) over the launch time — a MODEL rate, not HBM traffic. Where that model exceeds the peak "
                             "(the sketch lines, the many-flow and hot-flow lines, this line on the fastest boxes) achieved / frac are set to a real "
                             "bound instead — the counter traffic when a fresh PMC file for this workload exists, else the records read once — "
                             "frac_basis says which, and the model stays under alg_model_GBs. frac_traffic and frac_stream_floor are always printed",
                # honest yardsticks next to SURVEY §8(d)'s algorithmic figure (which charges a slot read + write per record that
                # the LDS flow cache never performs): frac_stream_floor = the 144-byte records alone, read once, against the
                # peak; frac_traffic = HBM bytes the counters saw (roofline.traffic), against the peak
                "frac_stream_floor": round(144 * recs_per_launch / (ingest_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ingest_ms > 0 else None,
                "frac_traffic": None,
                "alg_bytes_per_record": alg_bytes, "records_per_launch": int(recs_per_launch),
                "launch_ms": round(ingest_ms, 4), "launches": int(st.ingest_launches),
                "lds_cache_hit_rate": round(1.0 - st.records_bypassed / max(1, n * (args.steps + args.warmup + (1 if dist_on else 0))), 4),
                "kernel_Mrecords_per_s": round(recs_per_launch / (ingest_ms * 1e-3) / 1e6, 1) if ingest_ms > 0 else None,
                "evict_launch_ms": round(st.evict_kernel_ms / max(st.evict_launches, 1), 4),
                "sketch_launch_ms": round(st.sketch_kernel_ms / max(st.sketch_launches, 1), 4) if st.sketch_launches else None,
            },
        }
        if dist_on:
            out["config"]["rccl_ranks"] = world if args.backend == "nccl" else 0
            out["config"]["backend"] = args.backend
            out["config"]["member_records_folded"] = records_folded
            if local_fold:
                out["config"]["unique_flows_total"] = keys_total
                out["config"]["exchange"] = {
                    "what": "rank 0, one instrumented step after the timed ones (host clock, device synchronised between the phases)",
                    **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in phase.items()},
                    "partial_bytes": PARTIAL_BYTES,
                    "routed_bytes_per_step_rank0": int(phase.get("partials_sent", 0)) * PARTIAL_BYTES,
                    "routed_bytes_if_records_were_routed": int(n * (world - 1) / world) * 144,
                    "sketch_allreduce_bytes": (2 * (4 << 20) * 8 + 2 * (1 << 14)) if sketches else 0,
                }
        ev_ms = st.evict_kernel_ms / max(st.evict_launches, 1)
        if ev_ms > 0 and not args.dedup and not local_fold:
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
                out["roofline"]["traffic_source"] = traffic_source(tf, tj)
                out["roofline"]["traffic_stale"] = out["roofline"]["traffic_source"]["traffic_stale"]
                if "roofline_evict" in out and tj.get("evict_traffic_bytes_per_call") and tj.get("evicted_flows") == out["roofline_evict"]["flows_per_launch"]:
                    out["roofline_evict"]["traffic"] = round(tj["evict_traffic_bytes_per_call"] / (ev_ms * 1e-3) / 1e9, 1)
                    out["roofline_evict"]["traffic_bytes_per_flow"] = round(tj["evict_traffic_bytes_per_call"] / tj["evicted_flows"], 1)
                break
        # ---- a model rate above the peak describes no bound: the line's fraction is that of a real one (round-5 review, item 3)
        rf = out["roofline"]
        frac_of_a_real_bound(rf)
        if sketch_sep_ms:
            rf["sketch_launch_in_launch_ms"] = True
        # ---- CPU baseline: the oracle (C restatement of pkg/flow.Accounter), 1 core, bounded sample
        if args.cpu_sample > 0 and not dist_on:     # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, d_recs, n, max_entries)
        # ---- the chip's copy bandwidth in this very run (state the peak used, confirm it with a D2D copy):
        # up to 4 GiB device-to-device (the stream's first half over its second: nothing reads the stream after this point),
        # read + written bytes over the time of the copy
        default_workload = (not dist_on and not args.dedup and not args.sketches and not args.hot_permille and args.variant == 0
                            and not args.chunk and not args.max_entries)
        if default_workload and not args.no_extras:
            # the extra legs regenerate streams into d_recs: before the copy test overwrites half of it
            for tb in tabs:
                tb.close()
            tabs = []
            tab = None
            try:
                out["extra"] = extras(args, torch, nf, synth, d_recs, d_out, gen_stream, n, keys)
            except Exception as exc:                      # never a reason to lose the bench line
                out["extra"] = {"error": repr(exc)[:300]}
        try:
            # The chip's own yardsticks, in this very run, from kernels shaped like the fold's — csrc/nfagg_synth.hip
            # y_read_records / y_read_stream / y_copy, HIP events around three passes each (round 5 timed torch.Tensor.copy_, a
            # library blit at 4.8 TB/s; the guide's float4 copy kernel does 6.3): the stream's first 4 GiB read as records (seven
            # 16-byte loads of every 144-byte record: whole lines cross the fabric), read as a plain 16-byte stream, and copied onto
            # its second half (nothing reads the stream after this point).
            nb = min(4 << 30, (n * 144) // 2 // 2304 * 2304)
            sink = torch.zeros(2, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            a_ptr, b_ptr = d_recs.data_ptr(), d_recs.data_ptr() + nb
            ms_rr = synth.yardstick(0, a_ptr, 0, nb, sink.data_ptr())
            ms_rs = synth.yardstick(1, a_ptr, 0, nb, sink.data_ptr())
            ms_cp = synth.yardstick(2, a_ptr, b_ptr, nb, sink.data_ptr())
            rf = out["roofline"]
            rf["hbm_read_records_measured_GBs"] = round(nb / (ms_rr * 1e-3) / 1e9, 1)
            rf["hbm_read_stream_measured_GBs"] = round(nb / (ms_rs * 1e-3) / 1e9, 1)
            rf["hbm_copy_measured_GBs"] = round(2 * nb / (ms_cp * 1e-3) / 1e9, 1)
            rf["hbm_yardsticks"] = "own kernels (csrc/nfagg_synth.hip y_*), %d bytes, HIP events, 3 passes after 1" % nb
            if rf.get("traffic"):
                # the call's counter traffic is ~95 % reads (pass 1 streams the records, pass 2 gathers the spilled ones): the
                # yardstick that matches its mix is the read stream, not the copy
                tr = rf["traffic"]
                rf["traffic_over_measured_read_stream"] = round(tr / rf["hbm_read_stream_measured_GBs"], 4)
                rf["traffic_over_measured_read_records"] = round(tr / rf["hbm_read_records_measured_GBs"], 4)
                rf["traffic_over_measured_copy"] = round(tr / rf["hbm_copy_measured_GBs"], 4)
        except Exception as exc:                      # a diagnostic, never a reason to lose the bench line
            out["roofline"]["hbm_copy_measured_GBs"] = None
            out["roofline"]["hbm_copy_error"] = str(exc)[:100]
        quiet.emit(json.dumps(out))
    for tb in tabs:
        tb.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _oracle():
    """The CPU oracle, for the cpu_baseline legs only (never the thing measured as the product, never shipped)."""
    from oracle import oracle as O
    O.build()
    return O


def cpu_baseline_small_table(host_records, max_entries):
    """cpu_baseline at the reference's default CACHE_MAX_FLOWS: the oracle Accounter (1 core) over the same records as
    extra.cache_max_flows_5000, evicting on full exactly as account.go:85-94 does."""
    O = _oracle()
    m2 = len(host_records)
    acc = O.Accounter(max_entries, 0)
    raw = host_records.view(np.uint8).reshape(-1)
    t0 = time.perf_counter()
    off, evs, cflows = 0, 0, 0
    while off < m2:
        off += acc.ingest(raw[off * 144:])
        if off < m2:
            cflows += len(acc.evict()); evs += 1
    cflows += len(acc.evict()); evs += 1
    dt = time.perf_counter() - t0
    acc.close()
    return {"ms": round(dt * 1e3, 2), "Mrecords_per_s": round(m2 / dt / 1e6, 1), "evictions": evs, "evicted_flows": int(cflows),
            "us_per_epoch": round(dt / evs * 1e6, 1), "kind": "port", "cores": 1,
            "what": "oracle/nfagg_oracle.c Accounter (C restatement of pkg/flow/account.go:58-124), same records, evict-on-full at %d" % max_entries}


def cpu_baseline(args, d_recs, n, max_entries):
    O = _oracle()
    m = min(args.cpu_sample, n)
    sample = d_recs[: m * 144].cpu().numpy()
    acc = O.Accounter(max_entries, 1 if args.dedup else 0)
    t1 = time.perf_counter()
    consumed = acc.ingest(sample)
    ev = acc.evict()
    cpu_dt = time.perf_counter() - t1
    acc.close()
    assert consumed == m
    res = {
        "value": round(m / cpu_dt / 1e6, 3), "unit": "Mrecords/s", "cores": 1, "kind": "port", "what": "C restatement of pkg/flow.Accounter (oracle/nfagg_oracle.c); the Go reference cannot be built here",
        "sample": "first %d records of rank 0's stream (%d flows), oracle Accounter ingest+evict, %.1f s" % (m, len(ev), cpu_dt),
        "host_cores_available": os.cpu_count(),
    }
    # best-effort multi-core variant of the same restatement: PARTITION, THEN FOLD (oracle/nfagg_oracle_mt.c) —
    # T threads hash and bucket their contiguous slices of the sample by key shard (indices, arrival order kept), then thread k
    # folds shard k through the same Accounter code. Both phases are timed; every record is looked at by one partitioner and one
    # folder. (Round 3 let every thread scan the whole sample and skip (T-1)/T of it.) The reference itself is one goroutine.
    T = max(2, min(64, (os.cpu_count() or 2) // 2))
    folded, mc_flows, part_s, fold_s, biggest = O.partition_fold_mt(sample, T, max_entries, 1 if args.dedup else 0)
    assert folded == m and mc_flows == len(ev), (folded, m, mc_flows, len(ev))
    res["multicore"] = {"value": round(m / (part_s + fold_s) / 1e6, 3), "unit": "Mrecords/s", "cores": T, "kind": "port",
                        "what": "partition-then-fold: %d threads bucket their slices by key shard, then fold one shard each (oracle/nfagg_oracle_mt.c)" % T,
                        "partition_s": round(part_s, 4), "fold_s": round(fold_s, 4),
                        "largest_shard_share": round(biggest, 4),      # a key lives in ONE shard: the hottest flow's shard bounds the fold
                        "sample": "same sample, %.2f s" % (part_s + fold_s)}
    # ... and the decomposition that is not bound by the hottest flow's shard — the GPU's own local fold on host cores
    # (oracle/nfagg_oracle_mt.c orc_local_fold_mt): every thread folds its contiguous slice into a table of its own, in arrival
    # order; then thread k merges key shard k's entries from the tables in slice order with the same AccumulateBase (it is its own
    # ordered merge of partials). Bit-exact against one Accounter: tests/test_oracle_mt.py. Accounter mode only.
    if not args.dedup:
        for threads in sorted({T, max(2, min(256, os.cpu_count() or 2))}):
            folded, lf_flows, fold_s, merge_s, share = O.local_fold_mt(sample, threads, max_entries)
            assert folded == m and lf_flows == len(ev), (folded, m, lf_flows, len(ev))
            key = "multicore_local_fold" if threads == T else "multicore_local_fold_all_cores"
            res[key] = {"value": round(m / (fold_s + merge_s) / 1e6, 3), "unit": "Mrecords/s", "cores": threads, "kind": "port",
                        "what": "local fold then key-sharded merge: %d threads fold their slices into tables of their own, then merge one key shard "
                                "each, slice order = arrival order (oracle/nfagg_oracle_mt.c orc_local_fold_mt)" % threads,
                        "fold_s": round(fold_s, 4), "merge_s": round(merge_s, 4), "largest_shard_share_of_merged_entries": round(share, 4),
                        "sample": "same sample, %.2f s" % (fold_s + merge_s)}
        # ... and with its threads PLACED (round-5 review: the unpinned figure varied 3.6 x by box and fell with more threads — the
        # scheduler's doing, not the algorithm's): thread t bound to the t-th CPU in NUMA-node order, so T threads fill one node
        # before the next is touched; the best of 64 / 128 / 256 threads is the CPU's figure
        try:
            O.mt_set_pinning(True)
            tried = {}
            for threads in sorted({t_ for t_ in (64, 128, 256) if t_ <= (os.cpu_count() or 2)} or {max(2, os.cpu_count() or 2)}):
                folded, lf_flows, fold_s, merge_s, share = O.local_fold_mt(sample, threads, max_entries)
                assert folded == m and lf_flows == len(ev), (folded, m, lf_flows, len(ev))
                tried[threads] = round(m / (fold_s + merge_s) / 1e6, 3)
            best = max(tried, key=tried.get)
            res["multicore_local_fold_pinned_best"] = {"value": tried[best], "unit": "Mrecords/s", "cores": best, "kind": "port",
                                                       "by_threads": {str(k): v for k, v in tried.items()},
                                                       "what": "the same local fold with thread t bound to the t-th CPU in NUMA-node order "
                                                               "(orc_mt_set_pinning): best of the thread counts tried"}
        except Exception as exc:
            res["multicore_local_fold_pinned_best"] = {"error": repr(exc)[:200]}
        finally:
            O.mt_set_pinning(False)
    return res


def leg_traffic(leg, records_per_call):
    """HBM bytes per ingest call of an extra leg, from the PMC passes of the same workload committed under profiles/ (newest file
    whose "leg" and records_per_call match; tools/profile_bench.sh + tools/summarize_prof.py). None when there is none."""
    import glob
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            tj = json.load(open(tf))
        except Exception:
            continue
        if tj.get("leg") == leg and tj.get("records_per_call") == int(records_per_call) and tj.get("traffic_bytes_per_call"):
            return int(tj["traffic_bytes_per_call"]), traffic_source(tf, tj)
    return None, None


_LIB_SHA = None


def loaded_lib_sha256():
    """sha256 of the libnfagg.so this process loaded: what a committed traffic measurement must have been taken on."""
    global _LIB_SHA
    if _LIB_SHA is None:
        import hashlib
        from netobserv_ebpf_agent_amd import _lib
        _LIB_SHA = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    return _LIB_SHA


def traffic_source(path, tj):
    """Where a traffic figure comes from, and whether it is STALE: taken on another build of the library than the one loaded now
    (tools/profile_bench.sh records the hash on the GPU box; files from before round 5 carry none and count as stale)."""
    return {"file": os.path.relpath(path, ROOT), "git_head": tj.get("git_head"), "lib_sha256": tj.get("lib_sha256"),
            "traffic_stale": tj.get("lib_sha256") != loaded_lib_sha256()}


def extras(args, torch, nf, synth, d_recs, d_out, gen_stream, n, keys):
    """The legs next to the headline, measured in the same run (bounded: a few seconds in all). Every block: what ran, ms per
    step, records/s, and — where one kernel family dominates — its HIP-event time per call against SURVEY §8(d)'s bytes and
    against the stream floor (the 144-byte records alone)."""
    ex = {}

    def device_leg(name, mode, sk, variant, hot, what, steps=3, n_keys=0, max_entries=DEFAULT_MAX_ENTRIES):
        try:                                          # a leg that fails says so under its own key; the others still run
            _device_leg(name, mode, sk, variant, hot, what, steps, n_keys, max_entries)
        except Exception as exc:
            ex[name] = {"what": what, "error": repr(exc)[:300]}

    def _device_leg(name, mode, sk, variant, hot, what, steps, n_keys, max_entries):
        if n_keys:                                   # another population than the headline's (thresholds are per population size)
            th_k = synth.zipf_thresholds(n_keys, args.zipf)
            d_th_k = torch.from_numpy(th_k.view(np.int64)).cuda()
            synth.stream_device(d_recs.data_ptr(), n, j0=0, seed=2, n_keys=n_keys, d_thresholds=d_th_k.data_ptr(), hot_permille=hot, variant=variant)
            torch.cuda.synchronize()
            del d_th_k
        else:
            gen_stream(variant, hot)
        cm_t = [torch.zeros(4 << 20, dtype=torch.int64, device="cuda") for _ in range(2)] if sk else None
        hll_t = [torch.zeros(1 << 14, dtype=torch.uint8, device="cuda") for _ in range(2)] if sk else None
        ext = [cm_t[0].data_ptr(), cm_t[1].data_ptr(), hll_t[0].data_ptr(), hll_t[1].data_ptr()] if sk else None
        torch.cuda.synchronize()
        with nf.FlowTable(max_entries=max_entries, device=torch.cuda.current_device(), sketches=(nf.SKETCH_CM | nf.SKETCH_HLL) if sk else 0,
                          profile=True, mode=mode, ext_sketch=ext) as tab:
            def one():
                rc, c = tab.ingest_device(d_recs.data_ptr(), n)
                assert rc == nf.OK and c == n, (rc, c)
                f = tab.evict_device(d_out.data_ptr(), (d_out.numel() - 16) // 144, nf.REASON_TIMEOUT)
                if sk:
                    tab.sketch_reset()
                return f
            one()
            tab.sync(); tab.reset_profile()
            t0 = time.perf_counter()
            for _ in range(steps):
                flows = one()
            tab.sync()
            dt = (time.perf_counter() - t0) / steps
            st = tab.stats()
        k_ms = (st.ingest_kernel_ms + st.sketch_kernel_ms) / max(st.ingest_launches, 1)
        alg = ALG_BYTES_INGEST + (ALG_BYTES_SKETCH if sk else 0)
        # Yardsticks. SURVEY §8(d)'s model (392 / 522 B per record: a slot read + write per record) is what the headline's
        # roofline.achieved uses; a design that folds in LDS never moves those bytes, so on these legs the model can exceed the
        # peak — it is reported as a RATE (alg_model_GBs), not as a fraction of a bound. The fractions are of things that are
        # bounds: the 144-byte records read once (frac_stream_floor) and the HBM bytes the counters saw (frac_traffic, from the
        # PMC passes of this leg under profiles/).
        tb, tsrc = leg_traffic(name, n)
        ex[name] = {"what": what, "steps": steps, "ms_per_step": round(dt * 1e3, 3), "Mrecords_per_s": round(n / dt / 1e6, 1),
                    "evicted_flows_per_step": int(flows), "ingest_call_ms": round(k_ms, 4),
                    "alg_bytes_per_record": alg,
                    "alg_model_GBs": round(alg * n / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                    "frac_stream_floor": round(144 * n / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else None,
                    "traffic_bytes_per_launch": tb, "traffic_source": tsrc,
                    "frac_traffic": round(tb / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (tb and k_ms > 0) else None}

    device_leg("configs2", nf.MODE_ACCOUNTER, True, 0, 0,
               "configs[2]: the configs[1] stream + Count-Min(d=4,w=2^20) + HLL(p=14) per src/dst IP, one ingest call + eviction per step, device-resident")
    device_leg("configs4_shape", nf.MODE_KERNEL_DEDUP, False, 2, 900,
               "configs[4] shape on one GPU: 90 % of the records one flow (two interfaces), NFAGG_MODE_KERNEL_DEDUP (bpf/flows.c:76-143 merge), "
               "one ingest call + eviction per step, device-resident")
    device_leg("dedup_zipf", nf.MODE_KERNEL_DEDUP, False, 2, 0,
               "the configs[1] stream with every flow seen on two interfaces (stream variant 2), NFAGG_MODE_KERNEL_DEDUP, one ingest call + "
               "eviction per step, device-resident")

    if n >= 50_000_000:
        # the many-flow regime: what configs[3]'s local fold puts every rank in (5.4 M of 10 M flows seen per rank at N = 8)
        d_big = torch.empty((10_000_000 + 4096) * 144 + 16, dtype=torch.uint8, device="cuda")
        keep_out = d_out
        d_out = d_big
        try:
            device_leg("flows_10m", nf.MODE_ACCOUNTER, False, 0, 0,
                       "the same number of records over 10 M unique flows (Zipf 1.1), hash-aggregate only, one ingest call + eviction per step, "
                       "device-resident: the regime of one rank of configs[3]", n_keys=10_000_000, max_entries=1 << 24)
        finally:
            d_out = keep_out
            del d_big

    # ---- host path (PCIe-inclusive): the route the cgo shim takes. 20 M records from pageable host memory through
    # nfagg_ingest (pinned double-buffered ring, H2D on its own stream) + nfagg_evict (records back to host memory)
    gen_stream(0, 0)
    m = min(20_000_000, n)
    host = d_recs[: m * 144].cpu().numpy().view(nf.FLOW_RECORD)
    h_flows = np.empty(keys, dtype=nf.FLOW_RECORD)                # the caller's eviction buffer, reused tick after tick
    h_flows.view(np.uint8)[::4096] = 0
    with nf.FlowTable(max_entries=DEFAULT_MAX_ENTRIES, device=torch.cuda.current_device()) as tab:
        def e2e():
            rc, c = tab.ingest(host)
            assert rc == nf.OK and c == m, (rc, c)
            return len(tab.evict(nf.REASON_TIMEOUT, out=h_flows))
        e2e()
        t0 = time.perf_counter()
        flows = e2e()
        dt = time.perf_counter() - t0
    ex["e2e"] = {"what": "host buffer (pageable) -> nfagg_ingest -> nfagg_evict to host memory: PCIe-inclusive, %d M records, 1 timed pass after 1 warm-up" % (m // 1_000_000),
                 "ms": round(dt * 1e3, 2), "Mrecords_per_s": round(m / dt / 1e6, 1), "GBs_host_to_device": round(m * 144 / dt / 1e9, 1),
                 "evicted_flows": int(flows), "bound": "PCIe Gen5 x16 + host memcpy into the pinned ring"}
    # the same with the caller's two buffers in page-locked memory (nfagg_host_alloc): DMA straight from / into them
    pin_in, pin_out = nf.PinnedRecords(m), nf.PinnedRecords(keys)
    pin_in.records[:] = host
    with nf.FlowTable(max_entries=DEFAULT_MAX_ENTRIES, device=torch.cuda.current_device()) as tab:
        def e2e_pinned():
            rc, c = tab.ingest(pin_in.records)
            assert rc == nf.OK and c == m, (rc, c)
            return len(tab.evict(nf.REASON_TIMEOUT, out=pin_out.records))
        e2e_pinned()
        t0 = time.perf_counter()
        flows_p = e2e_pinned()
        dt = time.perf_counter() - t0
    assert flows_p == flows, (flows_p, flows)
    ex["e2e_page_locked"] = {"what": "the same from / into page-locked caller buffers (nfagg_host_alloc): no host copy into the staging ring",
                             "ms": round(dt * 1e3, 2), "Mrecords_per_s": round(m / dt / 1e6, 1), "GBs_host_to_device": round(m * 144 / dt / 1e9, 1),
                             "evicted_flows": int(flows_p), "bound": "PCIe Gen5 x16"}

    # ---- ring buffer -> pinned staging -> fold -> evict: the agent's own entry (pkg/flow/tracer_ringbuf.go:112-134 reads ONE sample,
    # decodes it with reflection and sends it down a channel; here nfagg_ringbuf_drain copies every committed sample of the BPF ring
    # straight into the buffer nfagg_staging_acquire lent out, nfagg_staging_commit sends it up and folds it while the next drain
    # runs). A synthetic producer refills a 64 MiB ring between the consumer's turns; only the consumer's time is counted.
    try:
        import ctypes as C
        Lb = nf._lib
        m_r = min(8_000_000, m)
        flat = np.zeros((m_r, 152), dtype=np.uint8)
        flat[:, 0] = 144                                                  # sample header: len = 144, flags 0, 4 bytes of padding
        flat[:, 8:] = host[:m_r].view(np.uint8).reshape(m_r, 144)
        flat = flat.reshape(-1)
        ring_bytes = 1 << 26
        ring = np.zeros(ring_bytes, dtype=np.uint8)
        prod, cons = np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64)
        rb = Lb.RingBuf(ring.ctypes.data, ring_bytes - 1, prod.ctypes.data, cons.ctypes.data)
        with nf.FlowTable(max_entries=DEFAULT_MAX_ENTRIES, device=torch.cuda.current_device()) as tab:
            def ring_pass():
                prod[0] = cons[0] = 152 * 1000 + 72                       # samples wrap around the data area somewhere
                k, t_cons, drains = 0, 0.0, 0
                n_c, sk_c, cap_c, buf_c, took = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_void_p(), C.c_size_t(0)
                while k < m_r or int(prod[0]) != int(cons[0]):
                    take = min(m_r - k, (ring_bytes - int(prod[0] - cons[0])) // 152)
                    if take:                                              # the kernel side (not timed)
                        pos, nb = int(prod[0]) & (ring_bytes - 1), take * 152
                        first = min(nb, ring_bytes - pos)
                        ring[pos:pos + first] = flat[k * 152:k * 152 + first]
                        ring[:nb - first] = flat[k * 152 + first:k * 152 + nb]
                        prod[0] += nb
                        k += take
                    t0 = time.perf_counter()
                    while True:                                           # the agent side
                        assert Lb.lib.nfagg_staging_acquire(tab._h, C.byref(buf_c), C.byref(cap_c)) == nf.OK
                        assert Lb.lib.nfagg_ringbuf_drain(C.byref(rb), buf_c, cap_c.value, C.byref(n_c), C.byref(sk_c), None) == nf.OK
                        assert Lb.lib.nfagg_staging_commit(tab._h, n_c.value, C.byref(took)) == nf.OK and took.value == n_c.value
                        drains += 1
                        if n_c.value < cap_c.value:
                            break
                    t_cons += time.perf_counter() - t0
                t0 = time.perf_counter()
                fl = len(tab.evict(nf.REASON_TIMEOUT, out=h_flows))
                return t_cons + time.perf_counter() - t0, fl, drains
            ring_pass()
            passes = sorted(ring_pass() for _ in range(7))               # seven timed passes: the spread is part of the result
            dt, fl, drains = passes[len(passes) // 2]
            rates = [round(m_r / p_[0] / 1e6, 1) for p_ in passes][::-1]
        ex["e2e_ring"] = {"Mrecords_per_s_min_median_max": [rates[0], rates[len(rates) // 2], rates[-1]], "passes": 7,
                          "host_copy_workers": nf.host_info(),
                          "what": "BPF-style ring buffer (64 MiB, refilled by a synthetic producer that is not timed) -> nfagg_staging_acquire -> "
                                  "nfagg_ringbuf_drain straight into the pinned staging buffer -> nfagg_staging_commit (H2D + fold, asynchronous) -> "
                                  "nfagg_evict to host memory: %d M records, consumer time only" % (m_r // 1_000_000),
                          "ms": round(dt * 1e3, 2), "Mrecords_per_s": round(m_r / dt / 1e6, 1), "drains": drains, "evicted_flows": int(fl),
                          "bound": "host cores copying 144-byte samples out of the ring into the pinned buffer (nfagg_ringbuf_drain: runs of plain samples over the copy workers, csrc/nfagg_hostpool.h), then PCIe"}
        del flat, ring
    except Exception as exc:
        ex["e2e_ring"] = {"error": repr(exc)[:300]}

    # ---- the reference's default CACHE_MAX_FLOWS = 5000 (pkg/config/config.go:146): the stream stops on "full" every few
    # thousand records. nfagg_account runs that loop on the device (one persistent kernel per staged chunk); next to it the
    # caller-driven loop of round 2 (nfagg_ingest -> NFAGG_FULL -> nfagg_evict per epoch)
    m2 = min(8_000_000, m)
    res = {}

    def account_leg(variant, into, M=5000):
        ends_cap = m2 // M + 16
        with nf.FlowTable(max_entries=M, device=torch.cuda.current_device(), ingest_variant=variant) as tab:
            d_ev = torch.empty((m2 + 2 * M + 8192) * 144, dtype=torch.uint8, device="cuda")
            h_ev = np.empty(m2 // 2 + 2 * M + 8192, dtype=nf.FLOW_RECORD)   # the caller's buffer for the evicted flows, reused call after call
            h_ev.view(np.uint8)[::4096] = 0                         # touched once: a long-lived buffer has its pages
            pin_ev = nf.PinnedRecords(m2 // 2 + 2 * M + 8192)
            # the closing eviction goes into a buffer the caller keeps too (a fresh array per call is page faults, not the library)
            h_close = np.empty(max(8192, M), dtype=nf.FLOW_RECORD)
            h_close.view(np.uint8)[::4096] = 0
            pin_close = nf.PinnedRecords(max(8192, M))
            def account(dev):
                if dev == 1:
                    rc, c, ends = tab.account_device(d_recs.data_ptr(), m2, d_ev.data_ptr(), m2 + 2 * M + 8192, ends_cap)
                    n_ep, flows = len(ends), (ends[-1] if ends else 0)
                else:
                    rc, c, epochs = (tab.account(host[:m2], out=h_ev, max_epochs=ends_cap) if dev == 0 else
                                     tab.account(pin_in.records[:m2], out=pin_ev.records, max_epochs=ends_cap))
                    n_ep, flows = len(epochs), sum(len(e) for e in epochs)
                assert rc == nf.OK and c == m2, (rc, c)
                flows += len(tab.evict(nf.REASON_CLOSING, out=h_close if dev == 0 else pin_close.records))
                return n_ep + 1, flows
            try:
                for dev in (0, 2, 1):
                    account(dev)
                    t0 = time.perf_counter()
                    evs, flows = account(dev)
                    dt = time.perf_counter() - t0
                    into[("account_host_path", "account_device_resident", "account_host_path_page_locked")[dev]] = {
                        "ms": round(dt * 1e3, 2), "Mrecords_per_s": round(m2 / dt / 1e6, 1), "evictions": evs, "evicted_flows": int(flows), "us_per_epoch": round(dt / evs * 1e6, 1)}
                st_ = tab.stats()
                into["paths"] = {"epochs_found_first": int(st_.account_epochs_first), "kernel_chain": int(st_.account_chain), "declined": int(st_.account_declined)}
            finally:
                del d_ev
                pin_ev.close(); pin_close.close()

    account_leg(0, res)
    # the same through the kernel chain alone (ingest_variant 30: what calls of a few epochs take, and the fallback of the default path)
    try:
        res["kernel_chain_variant_30"] = {}
        account_leg(30, res["kernel_chain_variant_30"])
    except Exception as exc:
        res["kernel_chain_variant_30"] = {"error": repr(exc)[:300]}
    tb5, tsrc5 = leg_traffic("cache_max_flows_5000", m2)           # the PMC passes of tools/account_5000_prof.py (same records, same call)
    if tb5 and "account_device_resident" in res:
        r5 = res["account_device_resident"]
        r5["traffic_bytes_per_launch"], r5["traffic_source"] = tb5, tsrc5
        r5["frac_traffic"] = round(tb5 / (r5["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)       # latency-bound: ~0.04
        r5["frac_stream_floor"] = round(144 * m2 / (r5["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    pin_out.close()                                              # (pin_in serves the legs below too)
    m3 = min(2_000_000, m2)
    with nf.FlowTable(max_entries=5000, device=torch.cuda.current_device()) as tab:
        def small(dev):
            off, evictions, flows = 0, 0, 0
            while off < m3:
                if dev:
                    rc, c = tab.ingest_device(d_recs.data_ptr() + off * 144, m3 - off)
                else:
                    rc, c = tab.ingest(host[off:m3])
                off += c
                if rc == nf.FULL:
                    flows += tab.evict_device(d_out.data_ptr(), 8192, nf.REASON_FULL) if dev else len(tab.evict(nf.REASON_FULL, cap=8192))
                    evictions += 1
            flows += len(tab.evict(nf.REASON_CLOSING, cap=8192))
            return evictions + 1, flows
        for dev in (False, True):
            small(dev)
            t0 = time.perf_counter()
            evs, flows = small(dev)
            dt = time.perf_counter() - t0
            res["ingest_evict_loop_device_resident" if dev else "ingest_evict_loop_host_path"] = {
                "ms": round(dt * 1e3, 2), "Mrecords_per_s": round(m3 / dt / 1e6, 1), "evictions": evs, "evicted_flows": int(flows), "us_per_epoch": round(dt / evs * 1e6, 1)}
    # the CPU side of the same leg (cpu_baseline_small_table, next to cpu_baseline): a 5000-entry map lives in the CPU's L1/L2 —
    # the CPU's best case, the GPU's worst (pkg/config/config.go:146 is the default)
    try:
        res["cpu_oracle_1_core"] = cpu_baseline_small_table(host[:m2], 5000)
    except Exception as exc:
        res["cpu_oracle_1_core"] = {"error": repr(exc)[:200]}
    ex["cache_max_flows_5000"] = {"what": "CACHE_MAX_FLOWS = 5000 (the reference's default), configs[1] stream, evict-on-full (account.go:85-94) every ~%d records: "
                                          "nfagg_account[_device] (%d M records; the loop runs on the device) and the caller-driven nfagg_ingest / nfagg_evict loop (%d M records)"
                                          % (m2 // max(res["account_host_path"]["evictions"], 1), m2 // 1_000_000, m3 // 1_000_000), **res}
    # ---- the table sizes the reference deploys and benchmarks: scripts/agent.yml:35-36 sets CACHE_MAX_FLOWS = 10 000,
    # pkg/flow/tracer_map_bench_test.go:64-111 brackets 1 k / 10 k / 100 k. Same records, same three routes.
    try:
        for M in (10_000, 100_000):
            r_ = {}
            try:
                account_leg(0, r_, M)
                tbm, tsrcm = leg_traffic("cache_max_flows_%d" % M, m2)          # the PMC passes of tools/account_5000_prof.py --max-entries M
                if tbm and "account_device_resident" in r_:
                    rm = r_["account_device_resident"]
                    rm["traffic_bytes_per_launch"], rm["traffic_source"] = tbm, tsrcm
                    rm["frac_traffic"] = round(tbm / (rm["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    rm["frac_stream_floor"] = round(144 * m2 / (rm["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                r_["cpu_oracle_1_core"] = cpu_baseline_small_table(host[:m2], M)
            except Exception as exc:
                r_["error"] = repr(exc)[:300]
            ex["cache_max_flows_%d" % M] = {"what": "CACHE_MAX_FLOWS = %d, configs[1] stream, %d M records per nfagg_account[_device] call: device-resident / "
                                                    "page-locked host buffers / pageable host buffers" % (M, m2 // 1_000_000), **r_}
        # ---- the calls the reference's own seam produces: Accounter.Account receives ONE record per channel operation through a
        # channel of BUFFERS_LENGTH = 50 (pkg/agent/agent.go:408, pkg/config/config.go:134, pkg/flow/tracer_ringbuf.go:112-134), so a
        # shim that hands over what is queued calls with 1 ... a few thousand records. Consecutive calls of n records over the same
        # stream (the map's state carries over), CACHE_MAX_FLOWS 5000; the one-core oracle on the same calls beside it.
        try:
            ex["shim_small_calls"] = small_calls(nf, host[:min(len(host), 4_000_000)], pin_in.records[:min(m2, 4_000_000)])
        except Exception as exc:
            ex["shim_small_calls"] = {"error": repr(exc)[:300]}
    finally:
        pin_in.close()
    return ex


def small_calls(nf, host, pinned, M=5000, sizes=(1, 64, 1024, 16384, 65536, 262144), max_calls=300):
    total = len(host)
    out_pin = nf.PinnedRecords(total // 2 + 2 * M + 8192)
    out_pg = np.empty(total // 2 + 2 * M + 8192, dtype=nf.FLOW_RECORD)
    out_pg.view(np.uint8)[::4096] = 0
    O = _oracle()
    raw = host.view(np.uint8).reshape(-1)
    rows = {}
    try:
        for n in sizes:
            calls = max(3, min(max_calls, total // n))
            row = {"calls": calls}
            for leg, src, dst in (("page_locked", pinned, out_pin.records), ("pageable", host, out_pg)):
                with nf.FlowTable(max_entries=M) as tab:
                    for k in range(min(3, calls)):                   # the first calls allocate (staging ring, scratch, graphs)
                        tab.account(src[k * n:(k + 1) * n], out=dst, max_epochs=n // M + 4)
                    tab.evict(nf.REASON_CLOSING, out=dst)
                    ts = []
                    t_all0 = time.perf_counter()
                    for k in range(calls):
                        t0 = time.perf_counter()
                        rc, c, _ = tab.account(src[k * n:(k + 1) * n], out=dst, max_epochs=n // M + 4)
                        ts.append(time.perf_counter() - t0)
                        assert rc == nf.OK and c == n, (rc, c)
                    t_all = time.perf_counter() - t_all0
                ts.sort()
                row[leg] = {"us_per_call_median": round(ts[len(ts) // 2] * 1e6, 1), "us_per_call_p90": round(ts[(len(ts) * 9) // 10] * 1e6, 1),
                            "Mrecords_per_s": round(n * calls / t_all / 1e6, 3)}
            acc = O.Accounter(M, 0)
            t_all0 = time.perf_counter()
            for k in range(calls):
                off, end = k * n, (k + 1) * n
                while off < end:
                    off += acc.ingest(raw[off * 144:end * 144])
                    if off < end:
                        acc.evict()
            t_all = time.perf_counter() - t_all0
            acc.close()
            row["oracle_1_core"] = {"us_per_call": round(t_all / calls * 1e6, 1), "Mrecords_per_s": round(n * calls / t_all / 1e6, 3)}
            rows[str(n)] = row
    finally:
        out_pin.close()
    # where a call starts to pay: the smallest size at which the page-locked call is faster than the one-core loop over the same records
    cross = next((int(k) for k, r in rows.items() if r["page_locked"]["us_per_call_median"] < r["oracle_1_core"]["us_per_call"]), None)
    return {"what": "consecutive nfagg_account calls of n records each, CACHE_MAX_FLOWS %d, from a page-locked and from a pageable buffer; "
                    "oracle_1_core: oracle/nfagg_oracle.c Accounter (C restatement of pkg/flow/account.go:58-124) over the same calls" % M,
            "by_call_size": rows, "first_size_faster_than_one_core": cross,
            "policy": "the shim gathers records until 65 536 wait or the oldest has waited 1 ms (netobserv-ebpf-agent_amd/accounter.py "
                      "BATCH_RECORDS / BATCH_TIMEOUT; INTEGRATION.md section 3)"}


def group_main(args, torch):
    """--group-devices: one process, D members, ONE common stream of D x --records records over D x --flows flows. Slice i of the
    stream (arrival order: slice 0 first) is resident on member i's device — as if it had come up that member's PCIe link — and
    enters through nfagg_group_ingest_device(i, ...): stable device partition by key-hash shard, buckets to their owners
    (hipMemcpyPeerAsync over xGMI between distinct devices), per-member folds; per step one sketch merge (RCCL all-reduce when
    the devices are distinct) and one eviction of every shard."""
    quiet = _QuietStdout()
    import __graft_entry__
    __graft_entry__.ensure_built()
    import netobserv_ebpf_agent_amd as nf
    from netobserv_ebpf_agent_amd import synth
    devs = [int(x) for x in args.group_devices.split(",")]
    n, keys1 = resolve_sizes(args, 1)
    D, keys = len(devs), keys1 * len(devs)
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
        outs.append(torch.empty((2 * keys1 + 4096) * 144, dtype=torch.uint8, device="cuda"))
    sk_flags = (nf.SKETCH_CM | nf.SKETCH_HLL) if args.sketches else 0
    max_entries = (args.max_entries or DEFAULT_MAX_ENTRIES) * D
    if args.group_local_fold:
        # max_entries stays D x the single-GPU table: it bounds every member undivided, and each member may see every flow
        outs = [torch.empty((2 * keys // D + 4096) * 144, dtype=torch.uint8, device="cuda:%d" % d) for d in devs]
    grp = nf.FlowGroup(devs, max_entries=max_entries, sketches=sk_flags, profile=True, local_fold=args.group_local_fold)
    out_cap = [o.numel() // 144 for o in outs]
    threads = args.group_threads

    def feed(i):
        rc, c = grp.ingest_device(i, slices[i].data_ptr(), n)
        assert rc == nf.OK and c == n, (rc, c)

    def step():
        if threads:
            import threading
            ths = [threading.Thread(target=feed, args=(i,)) for i in range(D)]
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
        else:
            for i in range(D):
                feed(i)
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
        "metric": METRIC,
        "value": round(n * D * steps / dt / 1e6, 3), "unit": "Mrecords/s", "n_gpus": len(set(devs)), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "configs[3] shape through nfagg_group_*: ONE common %dM-record Zipf(%.1f) stream over %dk flows, %d members on devices %s, "
                        "%s%s, device-resident input" % (n * D // 1_000_000, args.zipf, keys // 1000, D, devs,
                                                         "local fold, partials merged into their owners at the eviction" if args.group_local_fold
                                                         else "device partition + routing by key hash",
                                                         ", CM+HLL merged per step" if args.sketches else ""),
            "group_mode": "local_fold" if args.group_local_fold else "routed", "host_threads": D if threads else 1,
            "members": D, "devices": devs, "records_per_member_slice": n, "unique_flows_total": keys, "max_entries_total": max_entries,
            "evicted_flows_per_step": flows, "parallelism": "one process, group of %d members (distinct devices: %s)" % (D, len(set(devs)) == D),
            "member_fold_ms_per_launch": [round(x, 3) for x in fold_ms],
            "member_records_ingested": [int(st.records_ingested) for st in sts],
        },
    }
    quiet.emit(json.dumps(out))
    grp.close()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
