#!/bin/bash
# SQ counters of the fused SPADE -> 3x3 kernel and of the two launches it replaces, at the level-0 shape (two --pmc passes).
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/conv3_sq
RAW=/tmp/fsv_conv3_sq; mkdir -p "$OUT" $RAW; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $RAW/p$i -o p -- python "$ROOT/tools/spade_conv3_ab.py" --only ${SHAPE:-0} --reps 2 --rounds 1 > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($set): exit $?" | tee -a "$OUT/summary.txt"
done
python - $RAW <<'PY' | tee -a "$OUT/summary.txt"
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].strip()
        if not any(t in k for t in ('spade_conv3', 'spade_mod', 'conv_igemm_kernel<128, 32')):
            continue
        a = acc[k][r['Counter_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k[:70])
    for c, (n, t) in sorted(d.items()):
        print('   %-32s %14.0f per launch (%d launches)' % (c, t / n, n))
PY
