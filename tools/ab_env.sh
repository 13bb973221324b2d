#!/bin/bash
# same-box A/B of one environment switch on the whole training step: tools/ab_env.sh VAR=a VAR=b [out]
a=$1; b=$2; out=${3:-gpurun_out/ab_env.txt}
: > $out
for rep in 1 2; do
  for e in "$a" "$b"; do
    echo "== rep $rep $e" >> $out
    env $e timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inference --no-alt-mode 2>>$out.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f  ms %.2f  frac %.4f  losses %s' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['losses']))" >> $out || exit 1
  done
done
cat $out
