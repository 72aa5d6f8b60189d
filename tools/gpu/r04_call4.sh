#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04b
L=$PWD/netobserv-ebpf-agent_amd/lib
for lib in exp/libnfagg_expV1.so exp/libnfagg_expV2.so exp/libnfagg_expV3.so; do
 for v in "10 0" "10 1" "10 0 sf" "10 2 sf"; do
  echo "== $lib $v"
  NFAGG_LIB=$L/$lib timeout 120 python tests/tools/dedup_anatomy.py $v 2>&1 | grep -v "amdgpu.ids\|foreign key words" | tail -6
 done
done > gpurun_out/r04b/anatomy3.txt 2>&1
cat gpurun_out/r04b/anatomy3.txt
