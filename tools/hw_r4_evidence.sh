#!/bin/bash
# round 4 evidence: bench lines of the four workload / arithmetic combinations, rocprofv3 kernel statistics and PMC traffic of the
# headline step (pose fp32) and of BASELINE configs[4] in its stated arithmetic (street --amp O1), per-shape eager profiles
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r4ev}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash tools/hw_prof.sh ${TAG}_pose > /dev/null 2>&1
WARGS="--workload street --amp O1" PMC_META="street O1" bash tools/hw_prof.sh ${TAG}_street_amp > /dev/null 2>&1
cd "$ROOT"
for w in "street" "pose --amp O1"; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.tmp" 2> "$OUT/bench.err"
  tail -n 1 "$OUT/bench.tmp" >> "$OUT/bench_lines.jsonl"
done
tail -n 1 "gpurun_out/${TAG}_pose/bench_full.log" >> "$OUT/bench_lines.jsonl"
tail -n 1 "gpurun_out/${TAG}_street_amp/bench_full.log" >> "$OUT/bench_lines.jsonl"
timeout 300 python tools/shape_profile.py --workload street --amp O1 --top 70 --out "$OUT/shape_street_amp.jsonl" > "$OUT/shape_street_amp_top.txt" 2>&1
timeout 300 python tools/shape_profile.py --workload pose --top 70 --out "$OUT/shape_pose.jsonl" > "$OUT/shape_pose_top.txt" 2>&1
cut -c1-300 "$OUT/bench_lines.jsonl"
cat gpurun_out/${TAG}_pose/summary.txt gpurun_out/${TAG}_street_amp/summary.txt | cut -c1-260
