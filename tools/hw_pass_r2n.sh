#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2n
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
for rep in 1 2; do
run off_$rep        150 env FSV_BRANCH_STREAMS=0 $B
run flow_$rep       150 env FSV_BRANCH_TAGS=flow $B
run flow_refs_$rep  150 env FSV_BRANCH_TAGS=flow,refs $B
run flow_loss_$rep  150 env FSV_BRANCH_TAGS=flow,losses $B
run flow_emb_$rep   150 env FSV_BRANCH_TAGS=flow,emb $B
run refs_$rep       150 env FSV_BRANCH_TAGS=refs $B
run loss_$rep       150 env FSV_BRANCH_TAGS=losses $B
done
grep -o '"ms_per_step": [0-9.]*' "$OUT"/*.log
