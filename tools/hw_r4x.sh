#!/bin/bash
# round 4, second session: the early generator pass (model.early_generator) - graph tests on hardware, step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4x
mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python -m pytest tests/test_model_gpu.py -x -q -k "graph" > "$OUT/pytest_graph.txt" 2>&1
tail -n 5 "$OUT/pytest_graph.txt"
for f in 0 1 0 1; do
  FSV_EARLY_G=$f timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_$f.tmp" 2> "$OUT/bench_$f.err"
  echo "FSV_EARLY_G=$f $(tail -n 1 "$OUT/bench_$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
  tail -n 3 "$OUT/bench_$f.err"
done
for f in 0 1; do
  FSV_EARLY_G=$f timeout 200 python bench.py --workload street --amp O1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_s$f.tmp" 2> "$OUT/bench_s$f.err"
  echo "street amp FSV_EARLY_G=$f $(tail -n 1 "$OUT/bench_s$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
done
