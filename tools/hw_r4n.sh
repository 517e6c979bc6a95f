#!/bin/bash
# round 4, pass n: aten census at full size (street --amp, pose fp32)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4n}
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/aten_census.py --workload street --amp O1 --top 80 > "$OUT/aten_street_amp.txt" 2>&1
timeout 300 python tools/aten_census.py --workload pose --top 80 > "$OUT/aten_pose.txt" 2>&1
head -50 "$OUT/aten_street_amp.txt"
