#!/usr/bin/env python3
"""Does it matter on which NUMA node a page-locked caller buffer lives? The page-locked leg of nfagg_account (8 M records,
CACHE_MAX_FLOWS 5000) with the process bound to the CPUs of node 0 / node 1 while it allocates its buffers (first touch decides
where hipHostMalloc's pages are)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    node = int(sys.argv[1])
    cpus = set()
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        cpus |= set(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, cpus)
    sys.argv = [sys.argv[0], "--reps", "3"]
    sys.path.insert(0, ROOT)
    exec(open(os.path.join(ROOT, "tools", "account_paths_bench.py")).read())
else:
    sys.path.insert(0, ROOT)
    import netobserv_ebpf_agent_amd as nf
    print("GPU 0 hangs off NUMA node", nf.device_numa_node(0))
    for node in (0, 1, 0, 1):
        out = subprocess.run([sys.executable, __file__, str(node)], capture_output=True, text=True).stdout.strip().splitlines()
        j = json.loads([l for l in out if l.startswith("{")][-1])
        print("process on node", node, {k: (j[k]["ms_best"], j[k]["Mrecords_per_s"]) for k in ("device", "page_locked", "pageable")})
