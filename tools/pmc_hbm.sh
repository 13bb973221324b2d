#!/bin/bash
# HBM traffic of the HBM-bound 1x1 convs (res2 / res3 branch2c forward, res3 branch2a bwd-data): rocprofv3 --pmc FETCH_SIZE /
# WRITE_SIZE in separate passes (no trace domains) on tools/conv_bench.py.  Run on the GPU box from the repo root.
set -e
OUT=$PWD/gpurun_out/hbm_${1:-r01}
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/tools/conv_bench.py --shape res2c,res3c,res4c --iters 3 --mode fwd3,dgrad3 > $OUT/pmc_$C.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_* > gpurun_out/hbm_${1:-r01}.txt
python3 tools/conv_bench.py --shape res2c,res3c,res4c --iters 30 --mode fwd3,dgrad3 >> gpurun_out/hbm_${1:-r01}.txt
cat gpurun_out/hbm_${1:-r01}.txt
