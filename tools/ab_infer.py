"""same-box A/B of the config-3 inference leg (bench.inference_leg) under two environments: planes vs f32 storage"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, json, torch; sys.path.insert(0, %r); import numpy as np; import bench; bench.np = np; bench.torch = torch; "
        "from pyrapose_amd.runtime import default_context; r = bench.inference_leg(default_context(0), 'bf16x3'); "
        "print(json.dumps({k: r[k]['value'] for k in ('decode_compact', 'decode_compact_nms')}))" % ROOT)
for i in range(2):
    for env in ({"PP_PLANES": "1"}, {"PP_PLANES": "0"}):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
        print(env, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
