#!/bin/bash
# Round profile recipe (run on the GPU box from the repo root): kernel trace of the default bench, then PMC passes
# (one counter set per pass, no trace domains mixed in) on the dominant launch: the shared regression-head conv.
# Outputs under gpurun_out/prof_$1/ ; copy the summaries into profiles/.
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-mode --no-inference > $OUT/bench_traced.json 2> $OUT/bench_traced.err
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/tools/conv_bench.py --shape reg --iters 3 --fmt ${FMT:-1} --mode ${MODES:-fwd3pp,dgrad3pp,wgrad3p} > $OUT/pmc_$N.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
tail -1 $OUT/bench_traced.json | cut -c1-300
head -12 $OUT/kernel_stats.csv | cut -c1-160
cat $OUT/pmc_summary.txt
