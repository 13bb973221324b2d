#!/bin/bash
# the sparse (row-block list) launches of the 512-wide head conv in isolation: bwd-weight over listed blocks, bwd-data over listed tiles
export PP_SPLITK_MB=256 PP_PATCH=${PP_PATCH:-10}
run() { env "$@" python tools/conv_bench.py --shape reg --iters 20 --mode wgrad3sp,dgrad3sp 2>&1 | grep -v amdgpu | awk -v tag="$*" '{print tag, $1, $2, $7, $8}'; }
run PP_X=0
for s in 4 8 16 32 64; do run PP_WGRAD3_SP_STEPS=$s; done
for s in 4 8 12 24; do run PP_WGRAD3_SPLITS=$s; done
for s in 1 3; do run PP_SPARSE_DGRAD_SPLITS=$s; done
run PP_SPARSE_DGRAD=12
run PP_SPARSE_DGRAD=0
