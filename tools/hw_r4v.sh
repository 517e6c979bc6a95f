#!/bin/bash
# round 4, pass v: ordered split-K - operator test, the C1 step over seeds in the default mode (twice each), pose bench A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4v}
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_golden.py tests/test_model_gpu.py -q -m gpu -x > "$OUT/pytest.log" 2>&1
echo "tests: exit $? $(tail -n 2 "$OUT/pytest.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
cd tests
timeout 900 python - > "$OUT/c1_seeds.txt" 2>&1 <<'PY'
import torch, model_checks as mc
DEV = torch.device('cuda:0')
opt = mc.make_opt(dataset_mode='fewshot_face', input_nc=1, fineSize=128, loadSize=128, batchSize=1)
for seed in (21, 22, 23, 24):
    for rep in range(2):
        try:
            worst = mc.check_train_step(DEV, opt, b=1, tol=1e-3, grad_tol=1.0, seed=seed)
            print('seed', seed, 'rep', rep, 'worst grad rel L2 %.4e' % worst, flush=True)
        except AssertionError as e:
            print('seed', seed, 'rep', rep, 'FAILED', str(e)[:200], flush=True)
PY
cd "$ROOT"
cat "$OUT/c1_seeds.txt" | grep -v amdgpu | tee -a "$OUT/summary.txt"
for v in 1 0; do
  FSV_ORDERED_SPLIT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "pose fp32 ordered=$v: $(tail -n 1 "$OUT/bench.json" | cut -c1-200)" | tee -a "$OUT/summary.txt"
done
