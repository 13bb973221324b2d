#!/bin/bash
# wgrad3w ablations (PP_W3W_ABL bits: 1 = no fragment reads / MFMAs, 2 = no LDS stores, 4 = no global loads, 8 = no address walk;
# PP_W3W_EXP=1: a third of the x fragment reads), one process per variant, chained with && so that a variant that hangs ends the run
out=${1:-gpurun_out/r04_wgrad3w_abl.txt}
: > "$out"
run() { env PP_WGRAD3R=2 $1 timeout -k 5 40 python -u tools/conv_bench.py --fmt 1 --shape ${2:-reg} --mode wgrad3p --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/$1 /" >> "$out"; return ${PIPESTATUS[0]}; }
run PP_W3W_ABL=0 && run PP_W3W_ABL=1 && run PP_W3W_ABL=2 && run PP_W3W_ABL=3 && run PP_W3W_ABL=4
echo "rc=$?" >> "$out"
cat "$out"
