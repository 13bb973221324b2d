#!/bin/bash
# how much does the last, partly filled round of workgroups cost?  the 3x3 512->512 head conv over row counts that give
# 768 x {1, 2, 2.08 (the real launch), 2.5, 3} workgroups of 128x128 (126 output rows per tile, 4 column tiles)
out=${1:-gpurun_out/tail_probe.txt}
: > $out
for fmt in 1 0; do
  echo "== fmt $fmt" >> $out
  for H in 126 189 252 263 284 315 378; do   # x W=192: rows = 192 H -> ceil(192 H / 126) * 4 workgroups
    python tools/conv_bench.py --shape c:1:$H:192:512:512:3 --iters 20 --fmt $fmt --mode fwd3pp,dgrad3pp >> $out 2>&1 || exit 1
  done
done
cat $out
