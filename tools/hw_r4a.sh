#!/bin/bash
# round 4, pass a: the re-built N > 1 capture path on hardware (one-rank RCCL group + graphed-iteration checks), the headline after the
# housekeeping commit, and the C5 (street 1024x512) baseline in fp32 and with the round-2 narrow kernels (--amp O1) + its kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4a
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_rccl_gpu.py "tests/test_zz_np_gpu.py::test_graphed_iteration_on_hardware" -q -m gpu -x > "$OUT/pytest_gpu.log" 2>&1
echo "pytest: exit $? $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/bench_pose.json" 2> "$OUT/bench_pose.err"
echo "pose: $(tail -n 1 "$OUT/bench_pose.json" | cut -c1-220)" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --workload street --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_street_f32.json" 2> "$OUT/bench_street_f32.err"
echo "street f32: $(tail -n 1 "$OUT/bench_street_f32.json" | cut -c1-220)" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --workload street --amp O1 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_street_amp_old.json" 2> "$OUT/bench_street_amp_old.err"
echo "street amp (round-2 kernels): $(tail -n 1 "$OUT/bench_street_amp_old.json" | cut -c1-220)" | tee -a "$OUT/summary.txt"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_street -o p -- python "$ROOT/bench.py" --workload street --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/prof_street.log" 2>&1
for f in $(find /tmp/prof_street -name "*kernel_stats.csv"); do cp "$f" "$OUT/street_f32_kernel_stats.csv"; done
cd "$ROOT"
cat "$OUT/summary.txt"
