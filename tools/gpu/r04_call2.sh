#!/bin/bash
# round 4, GPU call 2: experiment — which build of the cached kernel-dedup path produces duplicate / missing flows
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04b
L=$PWD/netobserv-ebpf-agent_amd/lib
for v in "libnfagg.so 10 0" "exp/libnfagg_expA.so 10 0" "exp/libnfagg_expB.so 10 0" "libnfagg.so 10 0 sf" "libnfagg.so 1 0 sf" "libnfagg.so 10 1" "exp/libnfagg_expB.so 10 1"; do
  set -- $v
  echo "== $v"
  NFAGG_LIB=$L/$1 timeout 120 python tests/tools/dedup_anatomy.py $2 $3 $4 2>&1 | grep -v amdgpu.ids | tail -8
done > gpurun_out/r04b/anatomy.txt 2>&1
cat gpurun_out/r04b/anatomy.txt
