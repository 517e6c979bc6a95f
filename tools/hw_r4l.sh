#!/bin/bash
# round 4, pass l: per-shape eager timing of the street --amp step (SPADE f16 on / off)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4l}
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/shape_profile.py --workload street --amp O1 --top 90 --out "$OUT/shape_street_amp.jsonl" > "$OUT/shape_street_amp.txt" 2>&1
FSV_SPADE_F16=0 timeout 300 python tools/shape_profile.py --workload street --amp O1 --top 90 > "$OUT/shape_street_amp_nof16.txt" 2>&1
grep spade "$OUT/shape_street_amp.txt" | head -40
