#!/bin/bash
# same-box A/B of two builds of the library through the same Python: PP_LIB=<file under pyrapose_amd/> selects the .so
out=${1:-gpurun_out/ab_lib.txt}
: > $out
for rep in 1 2; do
  for lib in libpyrapose_hip_bf16x3.so libpyrapose_hip.so; do
    echo "== rep $rep $lib" >> $out
    PP_LIB=$lib timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inference 2>>$out.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f  ms %.2f  dense %.1f  sparse_fwd %s  frac %.4f  losses %s' % (r['value'], r['ms_per_step'], r['value_dense_backward'] or 0, (r['sparse_backward'].get('sparse_forward_opt_in') or {}).get('value'), r['roofline']['frac'], r['losses']))" >> $out || exit 1
  done
done
cat $out
