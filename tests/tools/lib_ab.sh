#!/bin/bash
# same-box A/B of two builds of libnfagg.so (NFAGG_LIB): lib/libnfagg_prev.so (built from the tree before a change) against lib/libnfagg.so
cd "$GRAFT_REPO_ROOT"
for rnd in 1 2; do
for lib in libnfagg_prev.so libnfagg.so; do
  echo "== $lib"
  NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/$lib timeout 300 python tests/tools/pass1_free_ab.py --variants 0 --reps 5 2>/dev/null | tail -1
  NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/$lib timeout 300 python tests/tools/pass1_free_ab.py --variants 0 --reps 3 --flows 10000000 2>/dev/null | tail -1
  if [ "${DEDUP:-0}" = "1" ]; then NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/$lib timeout 300 python tests/tools/pass1_free_ab.py --variants 0 --reps 3 --dedup 2>/dev/null | tail -1; fi
done
done
