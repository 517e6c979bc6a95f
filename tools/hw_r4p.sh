#!/bin/bash
# round 4, pass p: SPADE kernels as pixel-tile walkers
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4p}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_h_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "spade" > "$OUT/pytest_ops.log" 2>&1
echo "ops: exit $? $(tail -n 2 "$OUT/pytest_ops.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
for v in "" 0 1 2 4 6; do
  echo "== FSV_SPADE_WGS_PER_CU=$v f16" | tee -a "$OUT/spade_ab.txt"
  if [ -z "$v" ]; then timeout 300 python tools/spade_ab.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/spade_ab.txt"
  else FSV_SPADE_WGS_PER_CU=$v timeout 300 python tools/spade_ab.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/spade_ab.txt"; fi
done
echo "== fp32" | tee -a "$OUT/spade_ab.txt"
timeout 300 python tools/spade_ab.py --f16 0 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/spade_ab.txt"
echo "== fp32 one workgroup per tile" | tee -a "$OUT/spade_ab.txt"
FSV_SPADE_WGS_PER_CU=0 timeout 300 python tools/spade_ab.py --f16 0 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/spade_ab.txt"
for w in "street --amp O1" "street" "pose" "pose --amp O1"; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "$w: $(tail -n 1 "$OUT/bench.json" | cut -c1-260)" | tee -a "$OUT/summary.txt"
done
