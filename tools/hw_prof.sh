#!/bin/bash
# Default bench line + rocprofv3 kernel stats + the two PMC passes (FETCH_SIZE / WRITE_SIZE) of the bench command.  The raw
# rocprofv3 output stays in /tmp on the box; only the summaries are copied into gpurun_out/ (64 MiB merge limit).
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-prof}
RAW=/tmp/fsv_prof_raw_${1:-prof}
mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
run() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  local t0=$SECONDS
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? in $((SECONDS-t0))s ($(tail -n 1 "$OUT/$name.log" | cut -c1-200))" | tee -a "$OUT/summary.txt"
}
# WARGS: workload selection of every bench command (e.g. WARGS="--workload street --amp O1"; default: the headline pose fp32 step)
WARGS=${WARGS:-}
cd "$ROOT"
run bench_full   600 python bench.py $WARGS --steps 20 --warmup 5 --no-cpu-baseline --no-extras
cd /tmp
BARGS="$WARGS --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline"
run prof         300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/prof" -o p -- python "$ROOT/bench.py" $BARGS
run pmc_fetch    300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$RAW/pmc_fetch" -o p -- python "$ROOT/bench.py" $WARGS --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-roofline
run pmc_write    300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$RAW/pmc_write" -o p -- python "$ROOT/bench.py" $WARGS --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-roofline
cd "$ROOT"
find "$RAW" -type f | head -50 > "$OUT/raw_files.txt"; du -sh "$RAW"/* >> "$OUT/raw_files.txt" 2>&1
for f in $(find "$RAW/prof" -name "*stats*.csv"); do cp "$f" "$OUT/"; done
head -3 "$(find "$RAW/prof" -name "*kernel_trace.csv" | head -1)" > "$OUT/kernel_trace_head.csv"
python tools/trace_by_grid.py "$(find "$RAW/prof" -name "*kernel_trace.csv" | head -1)" --steps 13 --out "$OUT/trace_by_grid.jsonl" > "$OUT/trace_by_grid.log" 2>&1
python tools/step_sequence.py "$(find "$RAW/prof" -name "*kernel_trace.csv" | head -1)" --out "$OUT/step_sequence.txt" > "$OUT/step_sequence.log" 2>&1
python tools/pmc_traffic.py "$RAW/pmc_fetch" "$RAW/pmc_write" "$OUT/pmc_hbm_traffic.json" ${PMC_META:-} > "$OUT/pmc_summary.log" 2>&1
du -sh "$OUT" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
