#!/bin/bash
# round 3, pass m: slotted one-launch reductions (A/B against FSV_RED_SLOTS=0), float4 / four-loads-in-flight forms of the weight
# re-arrangement and gradient finalisation kernels (their durations from a kernel trace of the bench command)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r3m
RAW=/tmp/fsv_prof_raw
mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
cd "$ROOT"
timeout 600 python -m pytest tests -q -m gpu -x -k "layout_cache or deferred or reductions or test_norm or spade or conv_stats or adam" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest subset: exit $? $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
AB_NAME=r3m_ab REPS=2 bash tools/hw_ab.sh base slots0:FSV_RED_SLOTS=0 >> "$OUT/summary.txt" 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/prof" -o p -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > "$OUT/prof.log" 2>&1
echo "prof exit $?" | tee -a "$OUT/summary.txt"
cd "$ROOT"
for f in $(find "$RAW/prof" -name "*stats*.csv"); do cp "$f" "$OUT/"; done
python tools/trace_by_grid.py "$(find "$RAW/prof" -name "*kernel_trace.csv" | head -1)" --steps 13 --out "$OUT/trace_by_grid.jsonl" > "$OUT/trace_by_grid.log" 2>&1
cat "$OUT/summary.txt"
