#!/bin/bash
# where to release the next batch's frozen prefix: images/s of bench.py per PP_PREFETCH_AFTER (launch name), same box
out=${1:-gpurun_out/ab_prefetch}; mkdir -p $out
for round in 1 2; do
for v in OFF c2 res3d_branch2c res4f_branch2c res5c_branch2c reg_conv0 reg_conv2 reg_out; do
  if [ $v = OFF ]; then e="PP_PREFETCH=0"; elif [ $v = c2 ]; then e="PP_X=1"; else e="PP_PREFETCH_AFTER=$v"; fi
  env $e python bench.py --no-cpu-baseline --no-inference --no-alt-mode --steps 20 --warmup 5 > $out/$v.json 2>$out/$v.err || { tail -3 $out/$v.err; exit 1; }
  python -c "
import json;d=json.load(open('$out/$v.json'));print('%-16s %.1f img/s  %.3f ms  dense %.1f' % ('$v', d['value'], d['ms_per_step'], d['value_dense_backward'] or 0))"
done; done
