#!/bin/bash
# round 4, pass u: half side outputs of the element-wise producers (conversion passes removed)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4u}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_h_gpu.py tests/test_ops_gpu.py -q -m gpu -x > "$OUT/pytest.log" 2>&1
echo "tests: exit $? $(tail -n 2 "$OUT/pytest.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
for w in "street --amp O1" "pose --amp O1"; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "$w: $(tail -n 1 "$OUT/bench.json" | cut -c1-260)" | tee -a "$OUT/summary.txt"
  FSV_HALF_SIDE=0 timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "$w (no side outputs): $(tail -n 1 "$OUT/bench.json" | cut -c1-260)" | tee -a "$OUT/summary.txt"
done
cd /tmp
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_street -o p -- python "$ROOT/bench.py" --workload street --amp O1 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/prof_street.log" 2>&1
for f in $(find /tmp/prof_street -name "*kernel_stats.csv"); do cp "$f" "$OUT/street_amp_kernel_stats.csv"; done
grep -E "cast_f2h|norm_apply4|norm_bwd_apply4|act_bwd" "$OUT/street_amp_kernel_stats.csv" | cut -c1-200
