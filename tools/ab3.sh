#!/bin/bash
# like tools/ab.sh with three builds: build/ab/{old,new,new1}.so
set -e
for v in old new new1 old new new1; do
  cp build/ab/$v.so pyrapose_amd/libpyrapose_hip.so
  echo "== $v"
  if [ -n "$1" ]; then timeout -k 10 200 python3 tools/conv_bench.py $1; fi
  if [ "$2" = bench ]; then timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --no-alt-mode --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | cut -c1-200; fi
done
cp build/ab/old.so pyrapose_amd/libpyrapose_hip.so
