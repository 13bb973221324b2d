#!/bin/bash
# wgrad3r (tap-row reuse, csrc/conv4.hip) against wgrad3f in ONE process per format (PP_WGRAD3R is read at every launch)
out=${1:-gpurun_out/r04_wgrad_ab.txt}
mkdir -p "$(dirname "$out")"
{
  echo "== P16 (heads / FPN): PP_WGRAD3R = 0 (wgrad3f) / 1 (wgrad3r); wgrad3p = dense, wgrad3sp = over the listed blocks of a sparse dy"
  timeout -k 10 400 python tools/conv_bench.py --fmt 1 --shape reg,reg0,regout,cls,mask --mode wgrad3p,wgrad3sp --ab PP_WGRAD3R=0,1 --iters 20 --check
  echo "== bf16 pairs (backbone 3x3)"
  timeout -k 10 400 python tools/conv_bench.py --fmt 0 --shape res3,res4,res5,cls --mode wgrad3p --ab PP_WGRAD3R=0,1 --iters 20
} > "$out" 2>&1
tail -3 "$out"
