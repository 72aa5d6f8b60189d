#!/bin/bash
cd "$GRAFT_REPO_ROOT"
L=$PWD/netobserv-ebpf-agent_amd/lib
for lib in libnfagg.so exp/libnfagg_r03.so libnfagg.so exp/libnfagg_r03.so; do echo "== $lib"; NFAGG_LIB=$L/$lib timeout 200 python tests/tools/small_table_phases.py 2>&1 | grep -v amdgpu; done
