#!/usr/bin/env python3
"""Test infrastructure: the seed-parametrised randomised GPU tests (account loop on the device, routed and local-fold groups,
ranks exchanging partials in both table modes, the map-merge join) driven with FRESH seeds for a time budget — the pytest
suite pins a handful of seeds each; this looks for the ones it does not. Usage: python tests/tools/soak_seeds.py [seconds] [first seed]"""
import os, sys, time, traceback
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import netobserv_ebpf_agent_amd as nf
from oracle import oracle as O
import test_account_gpu, test_group_gpu, test_group_local_fold_gpu, test_dedup_local_fold_gpu, test_partials_gpu, test_map_merge

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(seed0)


def map_merge(seed):
    n_pop = int(rng.choice([1, 10, 300, 5000]))
    test_map_merge.test_map_merge_matches_oracle(nf, O, seed, n_pop, int(rng.integers(0, n_pop + 1)), int(rng.integers(0, n_pop + 1)), int(rng.choice([1, 2, 4, 16])))


def group_routed(seed):
    r = np.random.default_rng(seed)                 # the test's own draws: skip the seeds whose shards fill every few records
    n_members = int(r.choice([1, 2, 3, 5, 8])); keys = int(r.choice([40, 2_000, 30_000])); n = int(r.choice([30_000, 120_000, 250_000]))
    max_entries = int(r.choice([n_members * 3, max(n_members, keys // 4), keys // 2 + 7, 1 << 20]))
    if max_entries < keys and n > 40_000 and max_entries >= keys // 8:
        return
    test_group_gpu.test_group_randomised_against_the_contract(nf, O, seed)


CASES = [
    ("account", lambda s: test_account_gpu.test_account_ragged_batches_hot_flows_and_sketches(nf, O, s)),
    ("group routed", group_routed),
    ("group local fold", lambda s: test_group_local_fold_gpu.test_local_fold_random_splits_and_epochs(nf, O, s)),
    ("group local fold, dedup", lambda s: test_dedup_local_fold_gpu.test_group_local_fold_random_splits_and_epochs(nf, O, s)),
    ("ranks, dedup", lambda s: test_dedup_local_fold_gpu.test_ranks_with_interleaved_ragged_chunks(nf, O, s)),
    ("ranks", lambda s: test_partials_gpu.test_ranks_with_interleaved_ragged_chunks(nf, O, s)),
    ("map merge", map_merge),
]
t_end = time.time() + budget
runs = {name: 0 for name, _ in CASES}
seed = seed0
while time.time() < t_end:
    for name, fn in CASES:
        seed += 1
        t0 = time.time()
        print("case", name, "seed", seed, end=" ... ", flush=True)
        try:
            fn(seed)
            print("%.1f s" % (time.time() - t0), flush=True)
        except Exception:
            print("FAILED:", name, "seed", seed, flush=True)
            traceback.print_exc()
            sys.exit(1)
        runs[name] += 1
print("soak_seeds ok:", runs, "seeds", seed0 + 1, "..", seed)
