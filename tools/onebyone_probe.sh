#!/bin/bash
# the backbone's 1x1 launches (plane in, plane out) by tile: time and bytes moved (in + out planes, 4 B per element)
out=${1:-gpurun_out/onebyone_probe.txt}
: > $out
for tile in default "2,2" "1,2" "2,1" "1,1"; do
  echo "== tile $tile" >> $out
  if [ "$tile" = default ]; then unset PP_CONV3_TILE; else export PP_CONV3_TILE=$tile; fi
  PP_CONV_DEBUG=1 python tools/conv_bench.py --shape c:8:30:40:256:1024:1,c:8:30:40:1024:256:1,c:8:60:80:128:512:1,c:8:60:80:512:128:1,c:8:15:20:512:2048:1 --iters 30 --fmt 0 --mode fwd3pp,fwd3pp,dgrad3pp 2>&1 | grep -v "amdgpu.ids" | grep -v "^wgrad3\|splits 1 (may" | sort -u >> $out || exit 1
done
cat $out
