for shape in res4c res5c; do
for tile in 2,2 1,2 2,1 1,1; do
for sp in 2 4 8 16 32; do
  r=$(PP_WGRAD3_TILE=$tile PP_WGRAD3_SPLITS=$sp python tools/conv_bench.py --shape $shape --mode wgrad3 --iters 30 2>&1 | grep -v amdgpu | awk '{print $7, $8, $9, $10}')
  echo "$shape tile=$tile splits=$sp $r"
done; done; done
