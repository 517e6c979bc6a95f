#!/bin/bash
# round 4, pass d: the --amp O1 step on the half-precision kernels: street (C5) and pose (C3) bench lines, kernel stats of the street step
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4d}
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
timeout 300 python bench.py --workload street --amp O1 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_street_amp.json" 2> "$OUT/bench_street_amp.err"
echo "street amp: $(tail -n 1 "$OUT/bench_street_amp.json" | cut -c1-1200)" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/bench_street_amp.err" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --workload street --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench_street_f32.json" 2> "$OUT/bench_street_f32.err"
echo "street f32: $(tail -n 1 "$OUT/bench_street_f32.json" | cut -c1-300)" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --amp O1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench_pose_amp.json" 2> "$OUT/bench_pose_amp.err"
echo "pose amp: $(tail -n 1 "$OUT/bench_pose_amp.json" | cut -c1-300)" | tee -a "$OUT/summary.txt"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_street -o p -- python "$ROOT/bench.py" --workload street --amp O1 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/prof_street.log" 2>&1
for f in $(find /tmp/prof_street -name "*kernel_stats.csv"); do cp "$f" "$OUT/street_amp_kernel_stats.csv"; done
cd "$ROOT"
