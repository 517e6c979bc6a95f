#!/bin/bash
# Round-2 hardware pass E: full-size parity tests, whole GPU suite, default bench line (CPU baseline + extras), rocprofv3
# kernel stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE) of the bench command.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2e
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
run pytest_full  900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu --durations=5
run bench_full   600 python bench.py --steps 20 --warmup 5
run tile_ab      120 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K1152" "M131072 N128 K576"
cd /tmp
run prof         300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o p -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline
run pmc_fetch    300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o p -- python "$ROOT/bench.py" --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-roofline
run pmc_write    300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o p -- python "$ROOT/bench.py" --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-roofline
cd "$ROOT"
python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_hbm_traffic.json" > "$OUT/pmc_summary.log" 2>&1
# keep the merge small: per-dispatch CSVs are large
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +20M -delete
run pytest_gpu   1200 python -m pytest tests -x -q -m gpu --deselect tests/test_fullsize_gpu.py --durations=8
cat "$OUT/summary.txt"
