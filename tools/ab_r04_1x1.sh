#!/bin/bash
# igemm4p (persistent LDS-DMA GEMM of the 1x1 convolutions) against igemm3f in ONE process per run (PP_CONV4P read per launch)
out=${1:-gpurun_out/r04_1x1_ab.txt}
mkdir -p "$(dirname "$out")"
{
  echo "== bf16 pairs: PP_CONV4P = 0 (igemm3f) / 1 (igemm4p, 4 stages)"
  timeout -k 10 300 python tools/conv_bench.py --fmt 0 --shape res5c,res5a,res4c,res4a,res3c,res3a,res2c,lat3 --mode fwd3pp,dgrad3pp --ab PP_CONV4P=0,1 --iters 30
  echo "== P16"
  timeout -k 10 300 python tools/conv_bench.py --fmt 1 --shape lat3,res4c --mode fwd3pp --ab PP_CONV4P=0,1 --iters 30
} > "$out" 2>&1
tail -3 "$out"
