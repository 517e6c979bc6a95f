#!/bin/bash
# round 3, second hardware pass: GPU suite with the rebuilt SPADE kernels / epilogue statistics / grouped launches, in-box A/B of
# the new switches, kernel trace of the default build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
NAME=${1:-r3b}
OUT=$ROOT/gpurun_out/$NAME
mkdir -p "$OUT"
cd "$ROOT"
t0=$SECONDS
timeout 1500 python -m pytest tests -q -m gpu -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest -m gpu: exit $? in $((SECONDS-t0))s: $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
AB_NAME=${NAME}_ab REPS=2 bash tools/hw_ab.sh ${AB_SPECS:-base nostats:FSV_CONV_STATS=0 nopair:FSV_SPADE_PAIR=0 nogroups:FSV_CONV_GROUPS=0} >> "$OUT/summary.txt" 2>&1
export TMPDIR=/tmp
RAW=/tmp/fsv_prof_raw; mkdir -p $RAW
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/prof" -o p -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > "$OUT/prof.log" 2>&1
cd "$ROOT"
for f in $(find "$RAW/prof" -name "*stats*.csv"); do cp "$f" "$OUT/"; done
python tools/trace_by_grid.py "$(find "$RAW/prof" -name "*kernel_trace.csv" | head -1)" --steps 13 --out "$OUT/trace_by_grid.jsonl" > "$OUT/trace_by_grid.log" 2>&1
cat "$OUT/summary.txt"
