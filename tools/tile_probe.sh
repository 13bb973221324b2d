#!/bin/bash
# 128x128 vs 256x128 tiles of the tap-row-reuse kernel in both plane formats (after f16c8 the loop is LDS-bound: fewer fragment reads per MFMA?)
out=${1:-gpurun_out/tile_probe.txt}
: > $out
for fmt in 1 0; do
  for tile in "2,2" "4,2"; do
    echo "== fmt $fmt tile $tile" >> $out
    PP_CONV3_TILE=$tile PP_CONV3_X4=1 python tools/conv_bench.py --shape reg,cls,c:1:378:192:512:512:3,c:1:508:192:512:512:3 --iters 20 --fmt $fmt --mode fwd3pp,dgrad3pp >> $out 2>&1 || exit 1
  done
done
cat $out
