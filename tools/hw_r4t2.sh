#!/bin/bash
# round 4, second session: two-piece backward at one GPU (FSV_BENCH_SPLIT) + the decoder stage's Adam / layouts next to the second
# piece (FSV_EARLY_ADAM)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4t2
mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python -m pytest tests/test_zz_np_gpu.py tests/test_rccl_gpu.py -x -q -k "graphed or rccl" > "$OUT/pytest_graph.txt" 2>&1
tail -n 4 "$OUT/pytest_graph.txt"
run() {
  echo "$1 $(env $1 timeout 200 python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>"$OUT/err.txt" | tail -n 1 | cut -c1-175)" | tee -a "$OUT/step_ab.txt"
  tail -n 2 "$OUT/err.txt" | grep -v amdgpu.ids
}
for r in 1 2; do
  run "FSV_BENCH_SPLIT=0" ""
  run "FSV_BENCH_SPLIT=1 FSV_EARLY_ADAM=0" ""
  run "FSV_BENCH_SPLIT=1 FSV_EARLY_ADAM=1" ""
done
run "FSV_BENCH_SPLIT=0" "--workload street"
run "FSV_BENCH_SPLIT=1 FSV_EARLY_ADAM=1" "--workload street"
run "FSV_BENCH_SPLIT=0" "--workload street --amp O1"
run "FSV_BENCH_SPLIT=1" "--workload street --amp O1"
