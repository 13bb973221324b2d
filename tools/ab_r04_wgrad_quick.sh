#!/bin/bash
out=${1:-gpurun_out/r04_wgrad_quick.txt}
{
  timeout -k 10 300 python tools/conv_bench.py --fmt 1 --shape reg,cls,mask --mode wgrad3p --ab PP_WGRAD3R=0,1 --iters 20
  timeout -k 10 300 python tools/conv_bench.py --fmt 0 --shape res4 --mode wgrad3p --ab PP_WGRAD3R=0,1 --iters 20
} > "$out" 2>&1
grep -v amdgpu.ids "$out"
