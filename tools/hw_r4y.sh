#!/bin/bash
# round 4, second session: XCD bands in the fp32 gather-GEMM (FSV_CONV_BAND) - step A/B and per-shape profiles
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4y
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "test_conv" > "$OUT/pytest_conv.txt" 2>&1
tail -n 3 "$OUT/pytest_conv.txt"
for f in 0 1 0 1; do
  FSV_CONV_BAND=$f timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_$f.tmp" 2> "$OUT/bench_$f.err"
  echo "FSV_CONV_BAND=$f $(tail -n 1 "$OUT/bench_$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
done
for f in 0 1; do
  FSV_CONV_BAND=$f timeout 200 python bench.py --workload street --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_s$f.tmp" 2> "$OUT/bench_s$f.err"
  echo "street FSV_CONV_BAND=$f $(tail -n 1 "$OUT/bench_s$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
done
for f in 0 1; do
  FSV_CONV_BAND=$f timeout 300 python tools/shape_profile.py --workload pose --top 45 --out "$OUT/shape_pose_band$f.jsonl" > "$OUT/shape_pose_band${f}_top.txt" 2>&1
done
