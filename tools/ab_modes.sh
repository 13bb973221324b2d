#!/bin/bash
# same-box A/B of engine switches: tools/ab_modes.sh OUTDIR "ENV_A" "ENV_B" [rounds] -- alternates the two environments,
# prints images/s of bench.py (sparse default) for each run; use "X=1 Y=2" strings, "" for the default
out=$1; a=$2; b=$3; n=${4:-2}
mkdir -p $out
for i in $(seq 1 $n); do
  for tag in A B; do
    if [ $tag = A ]; then e="$a"; else e="$b"; fi
    env $e python bench.py --no-cpu-baseline --no-inference --no-alt-mode --steps 20 --warmup 5 > $out/ab_${tag}_$i.json 2>/dev/null || exit 1
    python - <<PY
import json
d = json.load(open("$out/ab_${tag}_$i.json"))
print("$tag [$e] run $i: %.1f img/s  %.3f ms  conv kernels:" % (d["value"], d["ms_per_step"]), [(k["kernel"], round(k["avg_ms"], 4)) for k in d["roofline"]["per_kernel"]])
PY
  done
done
