#!/bin/bash
# Round-2 first hardware pass (trimmed tools/first_hw_pass.sh): which opt-in paths are right / faster on a real MI355X.
set -u
OUT=gpurun_out/r2a
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  local t0=$SECONDS
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? in $((SECONDS-t0))s ($(tail -n 1 "$OUT/$name.log" | cut -c1-300))" | tee -a "$OUT/summary.txt"
}
run pytest_new      600 python -m pytest tests/test_zz_np_gpu.py -q -m gpu -rxX
run bench_f32       200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run bench_pf2       150 env FSV_TILE_REMAP=4:16,9:17,1:18 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_xcd       150 env FSV_TILE_REMAP=4:19,9:20,1:21 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_db        150 env FSV_TILE_REMAP=4:13,9:14,1:15 FSV_WGRAD_VARIANT=db python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_fusedfin  150 env FSV_FUSED_FINAL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_dgradmrg  150 env FSV_DGRAD_MERGE=2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_splitws   150 env FSV_SPLITK_WS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_allopt    150 env FSV_SPLITK_WS=2 FSV_DGRAD_MERGE=2 FSV_FUSED_FINAL=1 FSV_TILE_REMAP=4:16,9:17,1:18 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run tile_ab         240 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K1152" "M2048 N512 K2304" "M131072 N128 K576" "M512 N1024 K4608"
run wgrad_ab        240 python tools/wgrad_ab.py
run bench_amp_o1    150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --amp O1
run bench_bf16x3    150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --amp bf16x3
run bench_f32_again 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
cat "$OUT/summary.txt"
