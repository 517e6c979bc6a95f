#!/bin/bash
# round 3, first hardware pass: the whole GPU suite with the new defaults (prefetch-distance-2 tiles, grouped launches), then
# in-box A/B of the two switches and a kernel trace of the default build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r3a
mkdir -p "$OUT"
cd "$ROOT"
t0=$SECONDS
timeout 1200 python -m pytest tests -q -m gpu -x -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest -m gpu: exit $? in $((SECONDS-t0))s: $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
AB_NAME=r3a_ab REPS=2 bash tools/hw_ab.sh base nogroups:FSV_CONV_GROUPS=0 nopf2:FSV_CONV_PF2=0 >> "$OUT/summary.txt" 2>&1
export TMPDIR=/tmp
RAW=/tmp/fsv_prof_raw; mkdir -p $RAW
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/prof" -o p -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > "$OUT/prof.log" 2>&1
cd "$ROOT"
for f in $(find "$RAW/prof" -name "*stats*.csv"); do cp "$f" "$OUT/"; done
python tools/trace_by_grid.py "$(find "$RAW/prof" -name "*kernel_trace.csv" | head -1)" --steps 13 --out "$OUT/trace_by_grid.jsonl" > "$OUT/trace_by_grid.log" 2>&1
tail -3 "$OUT/prof.log" | cut -c1-300 >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
