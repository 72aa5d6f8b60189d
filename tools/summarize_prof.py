#!/usr/bin/env python3
"""Condense gpurun_out/prof (written by tools/profile_bench.sh on the GPU box) into
profiles/<tag>_*: the rocprofv3 kernel-stats CSV as is, one markdown table with the
per-launch PMC averages of the nfagg kernels, and <tag>_traffic.json — HBM bytes per
ingest call from FETCH_SIZE/WRITE_SIZE, corrected with the factors measured in the SAME
session on tools/pmc_calib (kernels with exactly known byte counts in the ingest path's
access patterns), as MI355X_MICROARCH.md §HBM prescribes."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
leg = sys.argv[3] if len(sys.argv) > 3 else None          # round 4: the bench leg this profile belongs to (extra.<leg> looks it up)
os.makedirs("profiles", exist_ok=True)
ks = sorted(glob.glob(f"{src}/trace/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], f"profiles/{tag}_kernel_stats.csv")
for j in glob.glob(f"{src}/trace_bench.json"):
    shutil.copy(j, f"profiles/{tag}_bench_under_rocprof.json")


def collect(d, want=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)
    # gpurun merges into an existing gpurun_out/: keep only the newest run of this directory
    for f in sorted(files, key=os.path.getmtime)[-1:]:
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return agg


rows, per_kernel = [], collections.defaultdict(dict)
for d in sorted(glob.glob(f"{src}/pmc_*")):
    if not os.path.isdir(d):
        continue
    for k, v in collect(d).items():
        if "nfagg" not in k:
            continue
        name = k.split("(")[0].replace("void ", "")
        for c, x in v.items():
            rows.append((name, c, len(x), sum(x) / len(x)))
            per_kernel[name][c] = (len(x), sum(x))
# calibration: known bytes / counter KiB
KNOWN = 30000000 * 144
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in collect(f"{src}/calib_{c}").items():
        name = k.split("(")[0].replace("void ", "")
        if c in v and sum(v[c]) > 0:
            calib[(name, c)] = KNOWN / (sum(v[c]) / len(v[c]) * 1024.0)
with open(f"profiles/{tag}_pmc_summary.md", "w") as o:
    o.write(f"# rocprofv3 PMC averages per launch ({tag})\n\n")
    o.write("Collected by tools/profile_bench.sh: one `rocprofv3 --pmc <counters>` run per counter group, `bench.py --steps 1 --warmup 0`.\n")
    o.write("FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; calibration factors below.\n\n")
    o.write("| kernel | counter | launches | average per launch |\n|---|---|---|---|\n")
    for r in rows:
        o.write(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]:.1f} |\n")
    if calib:
        o.write("\n## Calibration (tools/pmc_calib, 4.32 GB per kernel, same session)\n\n")
        o.write("true bytes / (counter x 1024):\n\n| kernel | counter | factor |\n|---|---|---|\n")
        for (k, c), f in sorted(calib.items()):
            o.write(f"| {k} | {c} | {f:.3f} |\n")
print(open(f"profiles/{tag}_pmc_summary.md").read())
# traffic per ingest call = sum over the ingest kernels of (FETCH x f_read + WRITE x f_write)
f_read = calib.get(("calib_read_records", "FETCH_SIZE"))
f_write = calib.get(("calib_write_records", "WRITE_SIZE"))
if f_read and f_write:
    try:
        bench = json.load(open(f"{src}/pmc_FETCH_SIZE_bench.json"))
    except Exception:
        bench = None
    ingest = [k for k in per_kernel if any(s in k for s in ("k_fold", "k_pass1", "k_pass2", "k_merge_overflow", "k_ingest", "k_dedup_claim", "k_dedup_fold", "k_dedup_stream", "k_dedup_parts", "k_dedup_overflow", "k_finalize",
                                                                "k_sketch_update", "k_ep_", "k_par_", "radix_sort"))]
    if leg and leg.startswith("cache_max_flows_"):             # nfagg_account: the evictions are part of the call
        ingest += [k for k in per_kernel if "k_evict" in k and k not in ingest]
    evict = [k for k in per_kernel if "k_evict" in k]
    ev_calls = max((per_kernel[k].get("FETCH_SIZE", (1, 0))[0] for k in evict), default=1)
    ev_traffic = (sum(per_kernel[k].get("FETCH_SIZE", (0, 0))[1] for k in evict) * f_read +
                  sum(per_kernel[k].get("WRITE_SIZE", (0, 0))[1] for k in evict) * f_write) * 1024.0 / max(1, ev_calls)
    calls = max(1, bench["roofline"]["launches"]) if bench else 1
    fetch_kib = sum(per_kernel[k].get("FETCH_SIZE", (0, 0))[1] for k in ingest)
    write_kib = sum(per_kernel[k].get("WRITE_SIZE", (0, 0))[1] for k in ingest)
    traffic = (fetch_kib * f_read + write_kib * f_write) * 1024.0 / calls
    # the tree this was measured on: the library's hash as the GPU box saw it (tools/profile_bench.sh) and — summarised in the
    # build container, where .git is — the commit and whether the tree was clean
    import hashlib, subprocess
    try:
        lib_sha = open(f"{src}/lib_sha256.txt").read().strip()
    except Exception:
        lib_sha = None
    def _git(*a):
        try:
            return subprocess.check_output(["git", *a], text=True, stderr=subprocess.DEVNULL).strip()
        except Exception:
            return None
    local_lib = "netobserv-ebpf-agent_amd/lib/libnfagg.so"
    local_sha = hashlib.sha256(open(local_lib, "rb").read()).hexdigest() if os.path.exists(local_lib) else None
    out = {
        "tag": tag, "leg": leg, "ingest_calls": calls, "kernels": ingest,
        "lib_sha256": lib_sha or local_sha, "lib_sha256_source": "the GPU box" if lib_sha else "the build container at summary time",
        "git_head": _git("rev-parse", "HEAD"), "git_dirty_files": len((_git("status", "--porcelain") or "").splitlines()),
        "fetch_kib_per_call": fetch_kib / calls, "write_kib_per_call": write_kib / calls,
        "factor_read": f_read, "factor_write": f_write,
        "traffic_bytes_per_call": traffic,
        "workload": bench["config"]["workload"] if bench else None,
        "hot_permille": bench["config"].get("hot_permille") if bench else None,
        "stream_variant": bench["config"].get("stream_variant") if bench else None,
        "mode": bench["config"].get("mode") if bench else None,
        "max_entries": bench["config"].get("max_entries") if bench else None,
        "evict_traffic_bytes_per_call": ev_traffic if evict else None,
        "evicted_flows": bench["config"].get("evicted_flows_per_step") if bench else None,
        "records_per_call": bench["roofline"]["records_per_launch"] if bench else None,
        "note": "FETCH_SIZE/WRITE_SIZE summed over the ingest kernels of one nfagg_ingest_device call, each multiplied by the "
                "factor measured on tools/pmc_calib (lane-strided 144-byte record reads / 144-byte record writes).",
    }
    json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))
