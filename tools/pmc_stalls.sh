#!/bin/bash
# Where do the waves of the dominant kernels wait?  PMC passes (one counter set per pass, no trace domains) on the
# regression-head conv through tools/conv_bench.py.  Run on the GPU box from the repo root; summary -> gpurun_out/stalls_$1.txt
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/stalls_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for C in "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD" \
         "SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT" \
         "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$i -- python3 $ROOT/tools/conv_bench.py --shape ${SHAPE:-reg} --iters 3 --fmt ${FMT:-1} --mode ${MODES:-fwd3pp,wgrad3p} > $OUT/pmc_$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_* > gpurun_out/stalls_$TAG.txt
cat gpurun_out/stalls_$TAG.txt
