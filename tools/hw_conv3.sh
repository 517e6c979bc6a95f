#!/bin/bash
# One gpurun call for the fused SPADE -> 3x3 convolution kernel (csrc/spade_conv3.hip, round 6): its operator tests, the isolated A/B
# against the two launches, the bench workload's full-size parity test with the fusion switched on, and the whole step with / without.
#   tools/gpu.sh --timeout 1500 -- 'bash tools/hw_conv3.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/conv3
mkdir -p "$OUT"
cd "$ROOT"
t0=$SECONDS
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -rf -k "3x3 or shortcut" > "$OUT/pytest_ops.log" 2>&1
echo "operator tests: exit $? in $((SECONDS-t0))s: $(tail -n 1 "$OUT/pytest_ops.log")" | tee -a "$OUT/summary.txt"
timeout 600 python tools/spade_conv3_ab.py > "$OUT/ab.jsonl" 2> "$OUT/ab.err"
echo "isolated A/B: exit $?" | tee -a "$OUT/summary.txt"
cat "$OUT/ab.jsonl" | tee -a "$OUT/summary.txt"
t0=$SECONDS
FSV_SPADE_CONV3=1 timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -rf -k "c3_pose_512_b2_full_step or c5_street_1024x512_nc35_fp32" > "$OUT/pytest_full.log" 2>&1
echo "full-size parity with FSV_SPADE_CONV3=1: exit $? in $((SECONDS-t0))s: $(tail -n 1 "$OUT/pytest_full.log")" | tee -a "$OUT/summary.txt"
AB_NAME=conv3/step REPS=2 bash tools/hw_ab.sh two_launches fused:FSV_SPADE_CONV3=1 fused_rw32:FSV_SPADE_CONV3=1,FSV_S3_RW=32
cat "$OUT/step/summary.txt" >> "$OUT/summary.txt"
# HBM traffic of the two forms at the level-0 shape (separate --pmc passes, tools/pmc_traffic.py)
RAW=/tmp/fsv_conv3_raw; mkdir -p $RAW; export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/f -o p -- python "$ROOT/tools/spade_conv3_ab.py" --only 0 --reps 2 --rounds 1 > "$OUT/pmc_fetch.log" 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/w -o p -- python "$ROOT/tools/spade_conv3_ab.py" --only 0 --reps 2 --rounds 1 > "$OUT/pmc_write.log" 2>&1 )
python tools/pmc_traffic.py $RAW/f $RAW/w "$OUT/pmc_hbm_traffic_conv3.json" "pose level 0 conv_0" > "$OUT/pmc_summary.log" 2>&1
python - "$OUT/pmc_hbm_traffic_conv3.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict) and any(t in k for t in ('spade_conv3', 'spade_mod', 'conv_igemm')):
        print(k[:80], v)
PY
