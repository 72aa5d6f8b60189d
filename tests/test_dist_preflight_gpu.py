"""Multi-GPU preflight on the ONE GPU a test box has (BASELINE.json configs[3], configs[4]): everything the
N > 1 path of `bench.py --gpus N` calls, executed with backend "nccl" (= RCCL) and world size 1 — init_process_group with a
device id, the sketch all-reduces (int64 SUM for Count-Min, uint8 MAX for the HyperLogLog registers), all_to_all_single with split
sizes on device tensors, partials export / merge, evict_owned — so that the first real 8-GPU run is not the first time these
calls execute. And the process that holds BOTH RCCLs: torch's bundled librccl.so (torch.distributed) and the one libnfagg's group
API dlopens itself (csrc/nfagg_group.inc), side by side."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import assert_records_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NFAGG_BENCH_WATCHDOG="500")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("dedup", [False, True])
def test_bench_world_1_through_the_rccl_path(nf, dedup):
    """configs[3] (and, with --dedup, configs[4]) as one rank: the same step the scaling curve times."""
    n, keys = 600_000, 30_000
    j = _bench("--gpus", "1", "--force-dist", "--backend", "nccl", "--records", str(n), "--flows", str(keys), "--steps", "2", "--warmup", "1",
               *(("--dedup", "--hot-permille", "900") if dedup else ()))
    c = j["config"]
    assert j["n_gpus"] == 1 and c["rccl_ranks"] == 1 and c["backend"] == "nccl" and "REHEARSAL" not in c["parallelism"]
    assert ("configs[4]" if dedup else "configs[3]") in c["workload"] and "local fold" in c["parallelism"]
    assert c["member_records_folded"] == [3 * n]
    ex = c["exchange"]
    assert ex["partials_sent"] == 0 and ex["partials_received"] == 0           # one rank owns every flow: the all-to-all moves empty segments
    assert ex["all_to_all_ms"] > 0 and ex["sketch_allreduce_ms"] > 0 and ex["merge_evict_ms"] > 0
    from netobserv_ebpf_agent_amd import synth
    th = synth.zipf_thresholds(keys, 1.1)
    whole = synth.stream_host(n, seed=2, n_keys=keys, thresholds=th, hot_permille=900 if dedup else 0, variant=2 if dedup else 0)
    distinct = len(np.unique(np.ascontiguousarray(whole["id"]).view(np.uint8).reshape(len(whole), 40), axis=0))
    assert c["evicted_flows_per_step"] == distinct


def test_collectives_of_the_exchange_at_world_1(nf, O):
    """The individual calls with contents checked: uint8 MAX / int64 SUM all-reduces on the library's own sketch buffers, an
    all_to_all_single with explicit split sizes on device tensors, export -> exchange -> merge -> evict_owned."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        recs = O.gen_stream(200_000, seed=9, n_keys=8_000, thresholds=O.zipf_thresholds(8_000, 1.1), variant=1)
        cm_t = [torch.zeros(4 << 12, dtype=torch.int64, device="cuda") for _ in range(2)]
        hll_t = [torch.zeros(1 << 10, dtype=torch.uint8, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        ext = [cm_t[0].data_ptr(), cm_t[1].data_ptr(), hll_t[0].data_ptr(), hll_t[1].data_ptr()]
        with nf.FlowTable(max_entries=1 << 16, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_log2_width=12, hll_p=10, ext_sketch=ext, local_fold=True) as tab:
            tab.set_sequence(0)
            assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
            tab.sync()
            nf.distributed.merge_sketches(cm_t, hll_t)                      # all_reduce SUM (int64), MAX (uint8) through RCCL
            torch.cuda.synchronize()
            cs, cd, hs, hd = O.sketches(recs, 4, 12, 10)
            assert np.array_equal(cm_t[0].cpu().numpy().view(np.uint64), cs) and np.array_equal(hll_t[1].cpu().numpy(), hd)
            d_exp = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
            rc, counts, n_exp = tab.partials_export_device(1, 0, d_exp.data_ptr(), d_exp.numel() * 8 // tab.partial_bytes)
            assert rc == nf.OK and counts == [0]
            # the all-to-all of the exchange, with split sizes, on device tensors: rank 0 sends itself 1000 words and nothing else
            send = torch.arange(1000, dtype=torch.int64, device="cuda"); recv = torch.zeros(1000, dtype=torch.int64, device="cuda")
            dist.all_to_all_single(recv, send, [1000], [1000])
            torch.cuda.synchronize()
            assert torch.equal(recv, send)
            tab.partials_merge_device(1, 0, d_exp.data_ptr(), 0)
            d_out = torch.zeros((1 << 16) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            rc, flows = tab.evict_owned_device(1, 0, d_out.data_ptr(), 1 << 16, nf.REASON_TIMEOUT)
            assert rc == nf.OK
            got = nf.sort_by_key(d_out.cpu().numpy()[:flows * 144].view(nf.FLOW_RECORD))
        assert_records_equal(got, O.run_accounter(recs, 1 << 20)[0][1])
        # ---- both RCCLs in one process: libnfagg's group API loads librccl itself (dlopen) while torch's is initialised above
        with nf.FlowGroup([0], max_entries=1 << 16, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_log2_width=12, hll_p=10) as grp:
            assert grp.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
            grp.merge_sketches()                                              # ncclAllReduce through the library's own communicator
            assert np.array_equal(grp.members[0].sketch_snapshot(nf.CM_SRC), cs)
            assert np.array_equal(grp.members[0].sketch_snapshot(nf.HLL_SRC), hs)
            t = torch.ones(16, dtype=torch.int64, device="cuda")
            dist.all_reduce(t)                                                # and torch's still works next to it
            torch.cuda.synchronize()
            assert int(t.sum()) == 16
            assert_records_equal(nf.sort_by_key(grp.evict(nf.REASON_CLOSING)), O.run_accounter(recs, 1 << 20)[0][1])
    finally:
        dist.destroy_process_group()
