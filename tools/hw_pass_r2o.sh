#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2o
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
run a_flow_emb         150 env FSV_BRANCH_TAGS=flow,emb $B
run b_flow_emb_loss    150 env FSV_BRANCH_TAGS=flow,emb,losses $B
run c_flow_emb_refs1   150 env FSV_BRANCH_TAGS=flow,emb,refs1 $B
run d_all_refs1        150 env FSV_BRANCH_TAGS=flow,emb,losses,refs1 $B
run e_fel_q2           150 env FSV_BRANCH_TAGS=flow,emb,losses DEBUG_HIP_FORCE_GRAPH_QUEUES=2 $B
run f_fel_q8           150 env FSV_BRANCH_TAGS=flow,emb,losses DEBUG_HIP_FORCE_GRAPH_QUEUES=8 $B
run g_all_q8           150 env FSV_BRANCH_TAGS=flow,emb,losses,refs DEBUG_HIP_FORCE_GRAPH_QUEUES=8 $B
run h_off_nocap        150 env FSV_BRANCH_STREAMS=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $B
run i_fe_nocap         150 env FSV_BRANCH_TAGS=flow,emb DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $B
run j_off              150 env FSV_BRANCH_STREAMS=0 $B
run k_flow_emb         150 env FSV_BRANCH_TAGS=flow,emb $B
run l_off_batch        150 env FSV_BRANCH_STREAMS=0 DEBUG_HIP_GRAPH_BATCH_SIZE=4096 $B
grep -o '"ms_per_step": [0-9.]*' "$OUT"/*.log
