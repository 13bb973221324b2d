#!/bin/bash
# Round 4 kernel A/Bs in ONE process per shape set (tools/conv_bench.py --ab: the switches are read at every launch):
#   igemm4p (persistent LDS-DMA GEMM of the 1x1 convolutions) against igemm3f, backbone format (bf16 pairs);
#   igemm4x<NWM = 2> (128-row LDS-DMA tiles, two workgroups per CU) against igemm3x, head format (P16).
out=${1:-gpurun_out/r04_kernel_ab.txt}
mkdir -p "$(dirname "$out")"
{
  echo "== 1x1 convolutions, bf16 pairs: PP_CONV4P = 0 (igemm3f) / 1 (igemm4p, 4 stages)"
  timeout -k 10 300 python tools/conv_bench.py --fmt 0 --shape res5c,res5a,res4c,res4a,res3c,res3a,res2c,lat3 --mode fwd3pp,dgrad3pp --ab PP_CONV4P=0,1 --iters 30
  echo "== 1x1 convolutions, bf16 pairs: ring of 3 stages (PP_CONV4P_NST = 3 / 4)"
  timeout -k 10 300 python tools/conv_bench.py --fmt 0 --shape res5c,res4c,res3c,res2c --mode fwd3pp --ab PP_CONV4P_NST=3,4 --iters 30
  echo "== 3x3 convolutions of the 256-channel heads / FPN, P16: PP_CONV3_DMA2 = 0 (igemm3x) / 1 (igemm4x, 128-row tiles, 2 per CU)"
  timeout -k 10 300 python tools/conv_bench.py --fmt 1 --shape cls,mask,cls1r --mode fwd3pp,dgrad3pp --ab PP_CONV3_DMA2=0,1 --iters 30
  echo "== the same on bf16 pairs (res3 / res4 3x3 run there)"
  timeout -k 10 300 python tools/conv_bench.py --fmt 0 --shape cls,mask --mode fwd3pp --ab PP_CONV3_DMA2=0,1 --iters 30
} > "$out" 2>&1
tail -5 "$out"
