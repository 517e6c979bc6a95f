#!/bin/bash
# round 4, pass e: per-shape in-step profile of the street --amp O1 step
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4e}
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/shape_profile.py --workload street --amp O1 --top 70 --out "$OUT/shape_profile_street_amp.jsonl" > "$OUT/shape_profile_street_amp.txt" 2> "$OUT/shape.err"
tail -3 "$OUT/shape.err"
cat "$OUT/shape_profile_street_amp.txt"
