#!/bin/bash
# same-box A/B of environment settings on the whole training step: tools/ab_env3.sh out "VAR=a" "VAR=b X=y" ...  (two repetitions, interleaved)
out=$1; shift
mkdir -p "$(dirname "$out")"
: > $out
for rep in 1 2; do
  for e in "$@"; do
    echo "== rep $rep $e" >> $out
    env $e timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inference --no-alt-mode 2>>$out.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
sb=r.get('sparse_backward') or {}
print('value %.1f  ms %.2f  dense_backward %s  frac %.4f' % (r['value'], r['ms_per_step'], sb.get('dense_backward', r.get('value_dense_backward')), r['roofline']['frac']))" >> $out || exit 1
  done
done
cat $out
