#!/bin/bash
# same-box A/B of the arithmetics (engine.py conv_mode) through the same library: whole training step, 2 repetitions each
out=${1:-gpurun_out/ab_mode.txt}
: > $out
for rep in 1 2; do
  for mode in bf16x3 mixed f16c8; do
    echo "== rep $rep $mode" >> $out
    timeout -k 10 500 python bench.py --steps 20 --warmup 3 --conv-mode $mode --no-cpu-baseline --no-inference --no-alt-mode 2>>$out.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f  ms %.2f  frac %.4f  issue %.4f  losses %s' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['mfma_issue_frac'], r['losses']))" >> $out || exit 1
  done
done
cat $out
