#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / TCC hit of the head convs under an environment ("ENVS" = space-separated VAR=value list)
set -e
TAG=${1:-x}
OUT=$PWD/gpurun_out/traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/tr_$i -- python3 $ROOT/tools/conv_bench.py --shape ${SHAPE:-reg,cls} --iters 3 --fmt ${FMT:-1} --mode ${MODES:-fwd3pp,dgrad3pp} > $OUT/tr_$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/tr_* > gpurun_out/traffic_$TAG.txt
cat gpurun_out/traffic_$TAG.txt
