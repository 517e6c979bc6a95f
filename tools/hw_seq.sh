#!/bin/bash
# Ordered launch list of one replayed step (tools/step_sequence.py) for a list of environment variants:
#   tools/gpu.sh --timeout 900 -- 'bash tools/hw_seq.sh base nostreams:FSV_BRANCH_STREAMS=0'
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${SEQ_NAME:-seq}
mkdir -p "$OUT"
export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}
  envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  RAW=/tmp/fsv_seq_$name
  rm -rf "$RAW"; mkdir -p "$RAW"
  cd /tmp
  timeout 300 env $envs rocprofv3 --kernel-trace --output-format csv -d "$RAW" -o p -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > "$OUT/$name.log" 2>&1
  cd "$ROOT"
  python tools/step_sequence.py "$(find "$RAW" -name "*kernel_trace.csv" | head -1)" --out "$OUT/seq_$name.txt" >> "$OUT/$name.log" 2>&1
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/$name.log") $(head -1 "$OUT/seq_$name.txt")" | tee -a "$OUT/summary.txt"
done
