for shape in res4 res5 res4c res5a; do
for tile in 1,1 1,2 2,1 2,2; do
for sp in 1 2 4 8; do
  r=$(PP_SPLITK_MB=64 PP_CONV3_TILE=$tile PP_CONV3_SPLITS=$sp python tools/conv_bench.py --shape $shape --mode fwd3 --iters 30 2>&1 | grep -v amdgpu | awk '{print $8, $9, $10, $11}')
  echo "$shape tile=$tile splits=$sp $r"
done; done; done
