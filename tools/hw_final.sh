#!/bin/bash
# What the driver runs at round end, in one call: the whole `-m gpu` suite, smoke(), the default bench line; then the
# rocprofv3 / PMC profiles of the bench command (tools/hw_prof.sh).
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-final}
mkdir -p "$OUT"
cd "$ROOT"
t0=$SECONDS
timeout 1500 python -m pytest tests -q -m gpu -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest -m gpu: exit $? in $((SECONDS-t0))s: $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
t0=$SECONDS
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > "$OUT/smoke.log" 2>&1
echo "smoke: exit $? in $((SECONDS-t0))s: $(tail -n 1 "$OUT/smoke.log")" | tee -a "$OUT/summary.txt"
t0=$SECONDS
timeout 600 python bench.py > "$OUT/bench_default.log" 2>&1
echo "bench.py: exit $? in $((SECONDS-t0))s" | tee -a "$OUT/summary.txt"
tail -n 1 "$OUT/bench_default.log" | cut -c1-400 | tee -a "$OUT/summary.txt"
bash tools/hw_prof.sh "${1:-final}_prof" > /dev/null 2>&1
cat "$ROOT/gpurun_out/${1:-final}_prof/summary.txt" >> "$OUT/summary.txt"
