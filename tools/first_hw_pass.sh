#!/bin/bash
# One gpurun call that answers everything left open at the end of round 1 (code written after the GPU budget was spent):
#   1. are the narrow-operand (--amp) kernels and the experimental fp32 tiles right on the hardware?   (pytest, XPASS = yes)
#   2. what do the operand modes buy per layer shape?                                                 (tools/np_ab.py)
#   3. do the few-wave / double-buffered tiles beat the shipped ones anywhere?                          (tools/tile_ab.py, wgrad_ab.py)
#   4. whole step: fp32 (headline), the experimental tiles swapped in (FSV_TILE_REMAP), --amp O1, --amp bf16x3   (bench.py)
# Every step has its own timeout; results land in gpurun_out/first_hw/.  Usage:
#   /usr/local/graft/bin/gpurun --timeout 3600 -- 'bash tools/first_hw_pass.sh'
set -u
OUT=gpurun_out/first_hw
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? ($(tail -n 1 "$OUT/$name.log" | cut -c1-300))" | tee -a "$OUT/summary.txt"
}
run pytest_new      900 python -m pytest tests/test_zz_np_gpu.py -q -m gpu -rxX
run pytest_all     1200 python -m pytest tests -x -q -m gpu -rxX
run np_ab           240 python tools/np_ab.py
run tile_ab         300 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K1152" "M2048 N512 K2304" "M8192 N128 K512" "M32768 N64 K256" "M131072 N128 K576" "M512 N1024 K4608"
run wgrad_ab        400 python tools/wgrad_ab.py
run bench_f32       300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run bench_db        300 env FSV_TILE_REMAP=4:13,9:14,1:15 FSV_WGRAD_VARIANT=db python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_pf2       300 env FSV_TILE_REMAP=4:16,9:17,1:18 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_xcd       300 env FSV_TILE_REMAP=4:19,9:20,1:21 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_fewwave   300 env FSV_TILE_REMAP=4:10,9:11,1:12 FSV_WGRAD_VARIANT=fw python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_fusedfin  300 env FSV_FUSED_FINAL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_dgradmrg  300 env FSV_DGRAD_MERGE=2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_splitws   300 env FSV_SPLITK_WS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_allopt    300 env FSV_SPLITK_WS=2 FSV_DGRAD_MERGE=2 FSV_FUSED_FINAL=1 FSV_TILE_REMAP=4:16,9:17,1:18 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_amp_o1    300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --amp O1
run bench_bf16x3    300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --amp bf16x3
run bench_f32_again 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline      # drift check: same box, same code as bench_f32
cat "$OUT/summary.txt"
