#!/bin/bash
# round 4, pass q: paired half stores / loads in the SPADE kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4q}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_h_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "spade" > "$OUT/pytest_ops.log" 2>&1
echo "ops: exit $? $(tail -n 2 "$OUT/pytest_ops.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
timeout 300 python tools/spade_ab.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/spade_ab.txt"
for w in "street --amp O1" "street" "pose --amp O1"; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "$w: $(tail -n 1 "$OUT/bench.json" | cut -c1-260)" | tee -a "$OUT/summary.txt"
done
