#!/bin/bash
# same-box A/B of the f16c8 kernel variants against bf16x3 (box-to-box spread of one binary is up to 18 % on this kernel)
out=${1:-gpurun_out/ab_conv2.txt}
shapes=${2:-reg,cls}
: > $out
for rep in 1 2; do
  echo "== rep $rep" >> $out
  timeout -k 10 200 python tools/conv_bench.py --shape $shapes --mode fwd3p --check >> $out 2>&1 || exit 1
  for wr in 2 4; do
    echo "-- PP_CONV2_WR=$wr" >> $out
    PP_CONV2_WR=$wr timeout -k 10 200 python tools/conv_bench.py --shape $shapes --mode fwd2 --check >> $out 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $out
