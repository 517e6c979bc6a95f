#!/bin/bash
# round 4, second session: the G step's real-image discriminator pass behind the discriminator step on its side stream (FSV_EARLY_REAL)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4u2
mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python -m pytest tests/test_zz_np_gpu.py -x -q -k "graphed" > "$OUT/pytest_graph.txt" 2>&1
tail -n 4 "$OUT/pytest_graph.txt"
for f in 0 1 0 1; do
  FSV_EARLY_REAL=$f timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_$f.tmp" 2> "$OUT/bench_$f.err"
  echo "FSV_EARLY_REAL=$f $(tail -n 1 "$OUT/bench_$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
  tail -n 2 "$OUT/bench_$f.err"
done
for f in 0 1; do
  FSV_EARLY_REAL=$f timeout 200 python bench.py --workload street --amp O1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_s$f.tmp" 2> "$OUT/bench_s$f.err"
  echo "street amp FSV_EARLY_REAL=$f $(tail -n 1 "$OUT/bench_s$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
done
