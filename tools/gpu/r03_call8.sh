#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
echo "== pass 1, 1 Mi records per call"; timeout -k 5 100 python tools/phase_timing.py 1000000 1048576 8 2>&1 | grep -v amdgpu.ids
echo "== pass 2, 1 Mi records per call"; timeout -k 5 100 python tools/phase_timing.py 1000000 1048576 9 2>&1 | grep -v amdgpu.ids
echo "== pass 1, 100 M"; timeout -k 5 100 python tools/phase_timing.py 1000000 100000000 8 2>&1 | grep -v amdgpu.ids
