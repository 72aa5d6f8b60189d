"""Test infrastructure (uses the oracle as checker). Tiny GPU probe used while bringing the kernels up (run from the repo root): each case in its own table, with a watchdog."""
import faulthandler, sys, time
faulthandler.dump_traceback_later(150, exit=True)
sys.path.insert(0, ".")
import numpy as np
import netobserv_ebpf_agent_amd as nf
from oracle import oracle as O

def case(name, recs, max_entries, variant):
    t0 = time.time()
    try:
        with nf.FlowTable(max_entries=max_entries, ingest_variant=variant) as tab:
            rc, c = tab.ingest(recs.view(nf.FLOW_RECORD))
            n = len(tab)
            got = nf.sort_by_key(tab.evict(nf.REASON_CLOSING))
        want = O.run_accounter(recs, 1 << 30)[0][1]
        ok = got.tobytes() == want.tobytes()
        print(f"{name} variant={variant}: rc={rc} consumed={c} live={n} flows={len(got)} want={len(want)} exact={ok} {time.time()-t0:.2f}s", flush=True)
    except Exception as e:
        print(f"{name} variant={variant}: EXC {e} {time.time()-t0:.2f}s", flush=True)

for variant in (1, 0):
    case("1rec", O.gen_stream(1, seed=1, n_keys=5), 100, variant)
    case("64rec-1key", O.gen_stream(64, seed=1, n_keys=1), 100, variant)
    case("256rec-50keys", O.gen_stream(256, seed=1, n_keys=50, variant=1), 100, variant)
    case("10k-1kkeys", O.gen_stream(10000, seed=1, n_keys=1000, variant=1), 5000, variant)
    th = O.zipf_thresholds(100000, 1.1)
    case("2M-100kkeys", O.gen_stream(2000000, seed=2, n_keys=100000, thresholds=th), 1 << 20, variant)
