#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2l
mkdir -p "$OUT"
run() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  local t0=$SECONDS
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? in $((SECONDS-t0))s ($(tail -n 1 "$OUT/$name.log" | cut -c1-200))" | tee -a "$OUT/summary.txt"
}
cd "$ROOT"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline"
run bench_base   150 $B
run bench_branch 150 env FSV_BRANCH_STREAMS=1 $B
run bench_base2  150 $B
run bench_branch2 150 env FSV_BRANCH_STREAMS=1 $B
run pytest_branch 900 env FSV_BRANCH_STREAMS=1 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "pose_warp_combine or c3 or temporal" -rf
cat "$OUT/summary.txt"
