#!/bin/bash
# round 4, second session: the fused bn_s -> conv_s kernel (csrc/spade_conv.hip) on hardware - operator tests, isolated A/B, step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4w
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_with_the_shortcut or test_spade" > "$OUT/pytest_ops.txt" 2>&1
tail -n 5 "$OUT/pytest_ops.txt"
timeout 200 python tools/spade_conv_ab.py > "$OUT/spade_conv_ab.jsonl" 2> "$OUT/spade_conv_ab.err"
cat "$OUT/spade_conv_ab.jsonl"; tail -n 5 "$OUT/spade_conv_ab.err"
for f in 0 1 0 1; do
  FSV_SPADE_CONV_S=$f timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_$f.tmp" 2> "$OUT/bench_$f.err"
  echo "FSV_SPADE_CONV_S=$f $(tail -n 1 "$OUT/bench_$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
done
