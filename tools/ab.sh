#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box variance is 5-10 %): build/ab/old.so vs build/ab/new.so.
# usage: bash tools/ab.sh "<conv_bench args>" [bench]   (run on the GPU box from the repo root)
set -e
for v in old new old new; do
  cp build/ab/$v.so pyrapose_amd/libpyrapose_hip.so
  echo "== $v"
  if [ -n "$1" ]; then timeout -k 10 200 python3 tools/conv_bench.py $1; fi
  if [ "$2" = bench ]; then timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --no-alt-mode --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | cut -c1-200; fi
done
cp build/ab/new.so pyrapose_amd/libpyrapose_hip.so
