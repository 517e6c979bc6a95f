#!/bin/bash
# the whole hardware suite + smoke, as the driver runs them at round end
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/suite
mkdir -p "$OUT"
cd "$ROOT"
t0=$SECONDS
timeout 1500 python -m pytest tests/ -q -m gpu -rf --durations=25 > "$OUT/pytest_gpu.txt" 2>&1
echo "pytest -m gpu: exit $? in $((SECONDS-t0))s"
tail -n 15 "$OUT/pytest_gpu.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > "$OUT/smoke.txt" 2>&1
tail -n 3 "$OUT/smoke.txt"
