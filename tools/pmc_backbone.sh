#!/bin/bash
# MFMA utilisation of the backbone's convs (north star: "MFMA utilisation on backbone convs") and HBM / L2 traffic of the
# dominant head conv, planes-only storage: rocprofv3 --pmc passes (one counter set per pass, no trace domains) on
# tools/conv_bench.py.  Run on the GPU box from the repo root; summary -> gpurun_out/pmc_backbone_$1.txt
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/pmc_backbone_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for C in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/bb_$i -- python3 $ROOT/tools/conv_bench.py --shape res2b,res3,res4,res5,res3c,res4c,res5c,res5a --iters 3 --mode fwd3pp,dgrad3pp,wgrad3p > $OUT/bb_$i.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/tr_$i -- python3 $ROOT/tools/conv_bench.py --shape reg,cls --iters 3 --mode fwd3pp,dgrad3pp,wgrad3p > $OUT/tr_$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/bb_* $OUT/tr_* > gpurun_out/pmc_backbone_$TAG.txt
python3 tools/conv_bench.py --shape res2b,res3,res4,res5,res3c,res4c,res5c,res5a,reg,cls --iters 30 --mode fwd3pp,dgrad3pp,wgrad3p >> gpurun_out/pmc_backbone_$TAG.txt
tail -40 gpurun_out/pmc_backbone_$TAG.txt
