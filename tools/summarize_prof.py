#!/usr/bin/env python3
"""Condense gpurun_out/prof (written by tools/profile_bench.sh on the GPU box) into
profiles/<tag>_*: the rocprofv3 kernel-stats CSV as is, and one markdown table with the
per-launch PMC averages of the nfagg kernels."""
import collections
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(f"{src}/trace/**/*_kernel_stats.csv", recursive=True)
if ks:
    shutil.copy(ks[0], f"profiles/{tag}_kernel_stats.csv")
for j in glob.glob(f"{src}/trace_bench.json"):
    shutil.copy(j, f"profiles/{tag}_bench_under_rocprof.json")
rows = []
for d in sorted(glob.glob(f"{src}/pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            if "nfagg" not in k:
                continue
            name = k.split("(")[0].replace("void ", "")
            for c, x in v.items():
                rows.append((name, c, len(x), sum(x) / len(x)))
with open(f"profiles/{tag}_pmc_summary.md", "w") as o:
    o.write(f"# rocprofv3 PMC averages per launch ({tag})\n\n")
    o.write("Collected by tools/profile_bench.sh: one `rocprofv3 --pmc <counters>` run per counter group, `bench.py --steps 1 --warmup 0`.\n")
    o.write("FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them (see DESIGN.md for the gfx950 correction).\n\n")
    o.write("| kernel | counter | launches | average per launch |\n|---|---|---|---|\n")
    for r in rows:
        o.write(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]:.1f} |\n")
print(open(f"profiles/{tag}_pmc_summary.md").read())
