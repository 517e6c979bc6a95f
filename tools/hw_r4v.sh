#!/bin/bash
# round 4, second session: bias gradients of the SPADE backward from the twin's epilogue on the fp32 kernels (FSV_SPADE_DBSUM)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4v
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "spade" > "$OUT/pytest_ops.txt" 2>&1
tail -n 4 "$OUT/pytest_ops.txt"
for f in 0 1 0 1; do
  FSV_SPADE_DBSUM=$f timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_$f.tmp" 2> "$OUT/bench_$f.err"
  echo "FSV_SPADE_DBSUM=$f $(tail -n 1 "$OUT/bench_$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
done
for f in 0 1; do
  FSV_SPADE_DBSUM=$f timeout 200 python bench.py --workload street --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_s$f.tmp" 2> "$OUT/bench_s$f.err"
  echo "street FSV_SPADE_DBSUM=$f $(tail -n 1 "$OUT/bench_s$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
done
