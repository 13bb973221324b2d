#!/bin/bash
# same-box kernel-level A/B of library builds (PP_LIB): tools/ab_lib_conv.sh out "lib1 lib2 ..." shapes modes
out=${1:-gpurun_out/ab_lib_conv.txt}
libs=${2:-"libpyrapose_hip_bf16x3.so libpyrapose_hip.so"}
shapes=${3:-reg,cls}
modes=${4:-fwd3pp,dgrad3pp,wgrad3p}
: > $out
for rep in 1 2; do
  for lib in $libs; do
    echo "== rep $rep $lib" >> $out
    PP_LIB=$lib timeout -k 10 300 python tools/conv_bench.py --shape $shapes --mode $modes >> $out 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $out
