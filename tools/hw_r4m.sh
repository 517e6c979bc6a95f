#!/bin/bash
# round 4, pass m: epilogues that load before they store (SPADE backward twin, residual / mask operand of the conv kernels)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4m}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_h_gpu.py tests/test_ops_gpu.py -q -m gpu -x > "$OUT/pytest_ops.log" 2>&1
echo "ops: exit $? $(tail -n 2 "$OUT/pytest_ops.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
for w in "street --amp O1" "street" "pose" "pose --amp O1"; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "$w: $(tail -n 1 "$OUT/bench.json" | cut -c1-260)" | tee -a "$OUT/summary.txt"
done
timeout 300 python tools/shape_profile.py --workload street --amp O1 --top 90 > "$OUT/shape_street_amp.txt" 2>&1
grep "spade" "$OUT/shape_street_amp.txt" | head -8
