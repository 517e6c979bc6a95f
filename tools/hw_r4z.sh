#!/bin/bash
# round 4, second session: the f16 form of the fused bn_s -> conv_s kernel - operator tests, isolated A/B, street --amp step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4z
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_with_the_shortcut" > "$OUT/pytest_ops.txt" 2>&1
tail -n 8 "$OUT/pytest_ops.txt"
timeout 200 python tools/spade_conv_ab.py --amp 1 > "$OUT/spade_conv_ab_amp.jsonl" 2> "$OUT/spade_conv_ab_amp.err"
cat "$OUT/spade_conv_ab_amp.jsonl"; tail -n 5 "$OUT/spade_conv_ab_amp.err"
for f in 0 1 0 1; do
  FSV_SPADE_CONV_S=$f timeout 200 python bench.py --workload street --amp O1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_$f.tmp" 2> "$OUT/bench_$f.err"
  echo "street amp FSV_SPADE_CONV_S=$f $(tail -n 1 "$OUT/bench_$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
done
for f in 0 1; do
  FSV_SPADE_CONV_S=$f timeout 200 python bench.py --workload pose --amp O1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_p$f.tmp" 2> "$OUT/bench_p$f.err"
  echo "pose amp FSV_SPADE_CONV_S=$f $(tail -n 1 "$OUT/bench_p$f.tmp" | cut -c1-200)" | tee -a "$OUT/step_ab.txt"
done
