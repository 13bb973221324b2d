#!/bin/bash
out=${1:-gpurun_out/r04_wgrad3w_quick.txt}
{
  timeout -k 10 300 python tools/conv_bench.py --fmt 1 --shape reg,reg0,cls,mask --mode wgrad3p --ab PP_WGRAD3R=0,2 --iters 20
} > "$out" 2>&1
grep -v amdgpu.ids "$out"
