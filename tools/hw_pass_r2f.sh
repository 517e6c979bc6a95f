#!/bin/bash
# Round-2 hardware pass F: full-size parity (with failure details), default bench line, rocprofv3 stats + PMC passes, new tests.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2f
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  local t0=$SECONDS
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? in $((SECONDS-t0))s ($(tail -n 1 "$OUT/$name.log" | cut -c1-200))" | tee -a "$OUT/summary.txt"
}
cd "$ROOT"
run pytest_new   900 python -m pytest tests/test_rccl_gpu.py tests/test_ops_gpu.py tests/test_golden.py -q -m gpu -k "rccl or avgpool or gemm_tile or numD2 or fullwidth" -rfs
run bench_full   600 python bench.py --steps 20 --warmup 5
cd /tmp
run prof         300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o p -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline
run pmc_fetch    300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o p -- python "$ROOT/bench.py" --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-roofline
run pmc_write    300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o p -- python "$ROOT/bench.py" --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-roofline
cd "$ROOT"
python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_hbm_traffic.json" > "$OUT/pmc_summary.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
ls -R "$OUT" > "$OUT/files.txt"
run pytest_full  900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu --durations=5 -rf
cat "$OUT/summary.txt"
