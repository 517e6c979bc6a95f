#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2q
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
run a_base     150 $B
run b_ovl1     150 env FSV_BWD_OVERLAP=1 $B
run c_ovl2     150 env FSV_BWD_OVERLAP=2 $B
run d_base     150 $B
run e_ovl2     150 env FSV_BWD_OVERLAP=2 $B
run f_ovl2_nobranch 150 env FSV_BWD_OVERLAP=2 FSV_BRANCH_STREAMS=0 $B
run g_graph_ovl2 300 env FSV_BWD_OVERLAP=2 PYTHONPATH=tests python tests/graph_step_checks.py
grep -o '"ms_per_step": [0-9.]*' "$OUT"/*.log
