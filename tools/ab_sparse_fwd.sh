#!/bin/bash
# variants of the opt-in sparse forward (PP_SPARSE_FWD=1): lane of the mask head, split count of the listed-block launches
out=${1:-gpurun_out/ab_sparse_fwd}; mkdir -p $out
i=0
for e in "PP_X=0" "PP_SPARSE_FWD=1" "PP_SPARSE_FWD=1 PP_MASK_LANE=1" "PP_SPARSE_FWD=1 PP_SPARSE_DGRAD_SPLITS=1" "PP_SPARSE_FWD=1 PP_SPARSE_DGRAD_SPLITS=3" "PP_SPARSE_FWD=1 PP_SPARSE_DGRAD=12" "PP_SPARSE_FWD=1" "PP_X=0"; do
  i=$((i+1))
  env $e python bench.py --no-cpu-baseline --no-inference --no-alt-mode --steps 20 --warmup 5 > $out/v$i.json 2>$out/v$i.err || { tail -3 $out/v$i.err; exit 1; }
  python -c "
import json;d=json.load(open('$out/v$i.json'));print('%-50s %.1f img/s  %.3f ms' % ('$e', d['value'], d['ms_per_step']))"
done
