#!/usr/bin/env python3
"""Timeline of ONE nfagg_account_device call out of a rocprofv3 --kernel-trace csv (tools/gpu/r06_acc_timeline.sh): every kernel of the
last call of the run with its start (us from the call's first kernel), duration and queue, and the gaps nothing ran in.
usage: acc_timeline.py <kernel_trace.csv> [first kernel's name, default k_par_hash]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_par_hash"
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda x: x[0])
# the last call: from the last-but-(launches per call - 1) k_par_hash on; calls are separated by the closing k_evict<false> of a step
starts = [i for i, k in enumerate(ks) if first in k[2]]
ends = [i for i, k in enumerate(ks) if "k_reset_after_evict" in k[2]]
if not starts:
    sys.exit("no %s in the trace" % first)
# the last step = the kernels after the last-but-one closing eviction's reset
closing = [i for i in ends]
lo = 0
if len(closing) >= 2:
    # a step ends with a closing eviction; find the reset that precedes the last run of k_par_hash launches belonging to the last step
    last_hash = starts[-1]
    prev_resets = [i for i in closing if i < last_hash]
    # walk back over the resets of the same step (evictions on full inside the step) — the step's first hash follows the previous step's LAST reset
    step_hashes = [s for s in starts]
    # simple rule: take everything after the reset that precedes the first hash launched within 20 ms of the last one
    t_last = ks[last_hash][0]
    cand = [s for s in step_hashes if t_last - ks[s][0] < 20_000_000]
    lo = cand[0]
t0 = ks[lo][0]
busy_end = t0
print("%9s %9s %6s  %s" % ("start us", "dur us", "queue", "kernel"))
for s, e, name, q in ks[lo:]:
    gap = (s - busy_end) / 1e3
    if gap > 5:
        print("%9s %9.1f %6s  -- nothing running" % ("", gap, ""))
    short = name.split("(")[0].replace("nfagg::", "").replace("void ", "")
    if "rocprim" in short:
        short = "rocprim:" + ("onesweep" if "onesweep" in name else "histogram" if "histogram" in name else short[-40:])
    print("%9.1f %9.1f %6s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short[:70]))
    busy_end = max(busy_end, e)
print("call: %.1f us from the first kernel's start to the last kernel's end" % ((busy_end - t0) / 1e3))
