#!/bin/bash
# occupancy probe in P16: wave tiles 64x64 (3 workgroups / CU) against 64x32 / 32x64 / 32x32 (4 workgroups / CU, 16 waves) on a tail-free shape
out=${1:-gpurun_out/tile_probe2.txt}
: > $out
for tile in "2,2" "2,1" "1,2" "1,1"; do
  echo "== fmt 1 tile $tile" >> $out
  PP_CONV3_TILE=$tile python tools/conv_bench.py --shape c:1:378:192:512:512:3,reg --iters 20 --fmt 1 --mode fwd3pp,dgrad3pp >> $out 2>&1 || exit 1
done
cat $out
