#!/bin/bash
# rocprofv3 kernel trace of a bench command, aggregated per (kernel, grid): tools/hw_trace.sh <out name> <bench args...>
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
NAME=${1:-trace}; shift
OUT=$ROOT/gpurun_out/$NAME
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
STEPS=6; WARM=3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o p -- python "$ROOT/bench.py" "$@" --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline --no-extras > "$OUT/prof.log" 2>&1
for f in $(find /tmp/prof_$NAME -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats.csv"; done
cd "$ROOT"
python tools/trace_by_grid.py "$(find /tmp/prof_$NAME -name "*kernel_trace.csv" | head -1)" --steps $((STEPS + WARM)) --out "$OUT/trace_by_grid.jsonl" > "$OUT/trace_by_grid.log" 2>&1
tail -2 "$OUT/prof.log" | cut -c1-300
head -3 "$OUT/trace_by_grid.log"
