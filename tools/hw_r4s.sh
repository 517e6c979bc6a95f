#!/bin/bash
# round 4, pass s: the patch-resident half convolution kernel - parity on hardware, A/B against the gather form, the step
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4s}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_h_gpu.py -q -m gpu -x > "$OUT/pytest_h.log" 2>&1
echo "h: exit $? $(tail -n 2 "$OUT/pytest_h.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
FSV_HAB=fwd timeout 600 python tools/h_ab.py M8192 M32768 M131072 M524288 M2048 2>&1 | grep -v amdgpu.ids | tee "$OUT/h_ab_patch.jsonl" | cut -c1-700
for w in "street --amp O1" "pose --amp O1"; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "$w: $(tail -n 1 "$OUT/bench.json" | cut -c1-260)" | tee -a "$OUT/summary.txt"
  FSV_HCONV_PATCH=0 timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "$w (gather form only): $(tail -n 1 "$OUT/bench.json" | cut -c1-260)" | tee -a "$OUT/summary.txt"
done
