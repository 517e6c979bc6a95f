#!/bin/bash
# round 4, pass t: kernel statistics of the fp32 headline step (pose) and of the street fp32 step
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4t}
mkdir -p "$OUT"
export TMPDIR=/tmp
for w in pose street; do
  cd /tmp
  rm -rf /tmp/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o p -- python "$ROOT/bench.py" --workload $w --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/prof_$w.log" 2>&1
  for f in $(find /tmp/prof_$w -name "*kernel_stats.csv"); do cp "$f" "$OUT/${w}_f32_kernel_stats.csv"; done
done
