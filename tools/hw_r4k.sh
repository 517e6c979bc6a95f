#!/bin/bash
# round 4, pass k: f16 SPADE - operator checks on hardware, then the street / pose --amp bench lines and the kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4k}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_h_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "test_h_gpu or spade" > "$OUT/pytest_ops.log" 2>&1
echo "ops: exit $? $(tail -n 2 "$OUT/pytest_ops.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
bash tools/hw_r4d.sh "${1:-r4k}"
FSV_SPADE_F16=0 timeout 300 python bench.py --workload street --amp O1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench_street_amp_nof16.json" 2> "$OUT/bench_street_amp_nof16.err"
echo "street amp, fp32 SPADE GEMMs: $(tail -n 1 "$OUT/bench_street_amp_nof16.json" | cut -c1-300)" | tee -a "$OUT/summary.txt"
