#!/bin/bash
# Per-shape profile of the two measured steps (pose fp32, street --amp O1) on one stream + the default bench line.
#   tools/gpu.sh --timeout 1200 -- 'bash tools/hw_shapes.sh r06a'
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-shapes}
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/shape_profile.py --top 80 --out "$OUT/shape_profile_pose_one_stream.jsonl" > "$OUT/shape_profile_pose_one_stream_top.txt" 2>&1
echo "pose shapes: exit $?" | tee -a "$OUT/summary.txt"
timeout 300 python tools/shape_profile.py --workload street --amp O1 --top 80 --out "$OUT/shape_profile_street_amp.jsonl" > "$OUT/shape_profile_street_amp_top.txt" 2>&1
echo "street shapes: exit $?" | tee -a "$OUT/summary.txt"
timeout 300 python tools/aten_census.py > "$OUT/aten_census.txt" 2>&1
echo "census: exit $?" | tee -a "$OUT/summary.txt"
timeout 400 python bench.py --no-cpu-baseline > "$OUT/bench_default.log" 2>&1
echo "bench: exit $?" | tee -a "$OUT/summary.txt"
tail -n 1 "$OUT/bench_default.log" | cut -c1-600 | tee -a "$OUT/summary.txt"
