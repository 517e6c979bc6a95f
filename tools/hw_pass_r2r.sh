#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2r
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
run a_fuse0    150 env FSV_NORM_FUSE_MAX_MB=0 $B
run b_fuse16   150 $B
run c_fuse4    150 env FSV_NORM_FUSE_MAX_MB=4 $B
run d_fuse64   150 env FSV_NORM_FUSE_MAX_MB=64 $B
run e_fuse0    150 env FSV_NORM_FUSE_MAX_MB=0 $B
run f_fuse16   150 $B
run g_fuse1    150 env FSV_NORM_FUSE_MAX_MB=1 $B
run h_ops      600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -rf
run i_graph    300 env PYTHONPATH=tests python tests/graph_step_checks.py
run j_model    900 python -m pytest tests/test_model_gpu.py tests/test_golden.py -q -m gpu -rf
grep -o '"ms_per_step": [0-9.]*' "$OUT"/*.log
