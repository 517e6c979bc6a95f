#!/bin/bash
# round 4, pass o: SPADE kernels in isolation, parts switched off
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4o}
mkdir -p "$OUT"
cd "$ROOT"
for d in 0 1 2 4 7; do
  echo "== FSV_SPADE_DBG=$d" | tee -a "$OUT/spade_ab.txt"
  FSV_SPADE_DBG=$d timeout 300 python tools/spade_ab.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/spade_ab.txt"
done
