#!/bin/bash
# round 4, pass r: paired half stores / loads in the half-precision gather-GEMM epilogue
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4r}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_h_gpu.py -q -m gpu -x > "$OUT/pytest_ops.log" 2>&1
echo "h: exit $? $(tail -n 2 "$OUT/pytest_ops.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
for w in "street --amp O1" "street" "pose --amp O1" "pose"; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "$w: $(tail -n 1 "$OUT/bench.json" | cut -c1-260)" | tee -a "$OUT/summary.txt"
done
cd /tmp
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_street -o p -- python "$ROOT/bench.py" --workload street --amp O1 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/prof_street.log" 2>&1
for f in $(find /tmp/prof_street -name "*kernel_stats.csv"); do cp "$f" "$OUT/street_amp_kernel_stats.csv"; done
