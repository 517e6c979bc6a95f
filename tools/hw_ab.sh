#!/bin/bash
# In-box A/B of the bench step: every argument is `name` or `name:VAR=value,VAR=value`; each variant runs the bench command
# once per repetition (REPS, default 2), back to back on the same box - the only comparison that means anything, boxes differ
# by +-5 %.  Usage (on the GPU box, through gpurun):  bash tools/hw_ab.sh off:FSV_BRANCH_STREAMS=0 on
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${AB_NAME:-ab}
mkdir -p "$OUT"
cd "$ROOT"
B="python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras --no-roofline"
for rep in $(seq 1 "${REPS:-2}"); do
  for spec in "$@"; do
    name=${spec%%:*}
    envs=""
    if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
    timeout 200 env $envs $B > "$OUT/${name}_$rep.log" 2>&1
    echo "$name #$rep exit $? $(grep -o '"ms_per_step": [0-9.]*' "$OUT/${name}_$rep.log")" | tee -a "$OUT/summary.txt"
  done
done
