#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2s
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
run b_fuse1    150 $B
run c_fuse05   150 env FSV_NORM_FUSE_MAX_MB=0.5 $B
run d_fuse2    150 env FSV_NORM_FUSE_MAX_MB=2 $B
run e_fuse4    150 env FSV_NORM_FUSE_MAX_MB=4 $B
run f_fuse0    150 env FSV_NORM_FUSE_MAX_MB=0 $B
run g_fuse1    150 $B
run h_ops      600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -rf -k "fused or norm or spade"
run j_model    900 python -m pytest tests/test_model_gpu.py -q -m gpu -rf -k "pose_warp_combine or train_step_face or face_refinement"
grep -o '"ms_per_step": [0-9.]*' "$OUT"/*.log
