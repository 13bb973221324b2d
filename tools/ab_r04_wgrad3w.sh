#!/bin/bash
# wgrad3w (producer / consumer waves, csrc/conv4.hip) against wgrad3f (0) and wgrad3r (1), one process per format; GPU times
out=${1:-gpurun_out/r04_wgrad3w.txt}
{
  echo "== P16 (heads / FPN): PP_WGRAD3R = 0 (wgrad3f) / 1 (wgrad3r) / 2 (wgrad3w); wgrad3p = dense, wgrad3sp = listed blocks"
  timeout -k 10 400 python tools/conv_bench.py --fmt 1 --shape reg,reg0,regout,cls,mask --mode wgrad3p,wgrad3sp --ab PP_WGRAD3R=0,1,2 --iters 20 --check
  echo "== bf16 pairs (backbone 3x3)"
  timeout -k 10 400 python tools/conv_bench.py --fmt 0 --shape res3,res4,res5,cls --mode wgrad3p --ab PP_WGRAD3R=0,2 --iters 20
} > "$out" 2>&1
grep -v amdgpu.ids "$out"
