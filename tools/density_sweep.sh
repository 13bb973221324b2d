#!/bin/bash
# headline vs annotation density: value / dense-backward / executed share for K boxes per image
out=${1:-gpurun_out/density_sweep.jsonl}
: > $out
for k in 1 3 8 15; do
  timeout -k 10 400 python bench.py --steps 20 --warmup 3 --boxes-per-image $k --no-cpu-baseline --no-inference 2>gpurun_out/density_$k.err | tail -1 >> $out || exit 1
  echo "K=$k done"
done
python - <<'PY'
import json,sys
rows=[json.loads(l) for l in open("gpurun_out/density_sweep.jsonl") if l.strip().startswith("{")]
tab=[]
for r in rows:
    sh=r["sparse_backward"]["executed_share"]
    w=[v["wgrad"] for v in sh.values()]; d=[v["dgrad"] for v in sh.values()]
    tab.append({"boxes_per_image":r["config"]["boxes_per_image"],"value":r["value"],"ms_per_step":r["ms_per_step"],
                "value_dense_backward":r["value_dense_backward"],"sparse_forward_opt_in":(r["sparse_backward"].get("sparse_forward_opt_in") or {}).get("value"),
                "executed_share_wgrad_mean":sum(w)/len(w),"executed_share_dgrad_mean":sum(d)/len(d),"executed_share":sh,
                "roofline_frac":r["roofline"]["frac"],"workload":r["config"]["workload"]})
json.dump(tab,open("gpurun_out/density_sweep.json","w"),indent=1)
for t in tab: print(t["boxes_per_image"], round(t["value"],1), round(t["value_dense_backward"],1), round(t["executed_share_wgrad_mean"],3), round(t["executed_share_dgrad_mean"],3))
PY
