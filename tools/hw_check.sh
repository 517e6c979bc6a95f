#!/bin/bash
# One gpurun call = one checkpoint of the round: a named subset of the hardware suite, then bench lines.  Usage (through tools/gpu.sh):
#   tools/gpu.sh --timeout 1500 -- 'bash tools/hw_check.sh <tag> "<pytest selection>" [bench args ...]'
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-check}; SEL=${2:-}; shift 2 || true
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ -n "$SEL" ]; then
  t0=$SECONDS
  timeout 1200 python -m pytest $SEL -q -m gpu -rf -s > "$OUT/pytest.log" 2>&1
  echo "pytest [$SEL]: exit $? in $((SECONDS-t0))s: $(tail -n 1 "$OUT/pytest.log")" | tee -a "$OUT/summary.txt"
  grep -h "amp step vs\|FAILED\|Error" "$OUT/pytest.log" | head -20 | tee -a "$OUT/summary.txt"
fi
t0=$SECONDS
timeout 900 python bench.py "$@" > "$OUT/bench.log" 2> "$OUT/bench.err"
echo "bench.py $*: exit $? in $((SECONDS-t0))s" | tee -a "$OUT/summary.txt"
tail -n 1 "$OUT/bench.log" > "$OUT/bench.json"
python - "$OUT/bench.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get('roofline', {})
    print('value %s %s, %s ms/step, launch %s' % (d['value'], d['unit'], d['ms_per_step'], d['config'].get('launch')))
    print('roofline %s: %s us frac %s; in_timed_schedule %s; eager %s' % (r.get('kernel'), r.get('avg_launch_us'), r.get('frac'), r.get('in_timed_schedule'), r.get('eager_bracket_us')))
    ex = d.get('extras', {})
    for k, v in ex.items():
        print('extras', k, json.dumps(v)[:400])
    print('cpu_baseline', d.get('cpu_baseline'))
except Exception as e:
    print('no bench line:', e)
PY
tail -n 5 "$OUT/bench.err"
