"""profiles/rNN_traffic.json from the summaries tools/pmc_traffic.sh writes (one per plane format):
   python tools/traffic_json.py OUT.json mixed=gpurun_out/traffic_fmt1.txt bf16x3=gpurun_out/traffic_fmt0.txt [f32-from=profiles/r02_traffic.json]
FETCH_SIZE is reported in KiB and doubled (MI355X_MICROARCH.md: gfx950 tallies 128-byte requests as 64); WRITE_SIZE in KiB as reported.
The launch priced is the 3x3 512->512 head conv over P3|P4|P5 at batch 8 (grid 409600), forward / bwd-data on packed planes."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "pyrapose_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse(path):
    """counters of the regression-head launch: igemm4x_kernel (grid 768 x 512: the whole rounds, 48 768 of the 50 400 rows) where it
    ran, else igemm3x_kernel at grid 409 600 (all rows)"""
    v, v3 = {}, {}
    for line in open(path):
        m = re.match(r"(\S+<[^>]*>)\s+grid=(\d+)\s+(\S+)\s+n=\d+ mean=(\S+)", line)
        if not m:
            continue
        if "igemm4x" in m.group(1) and m.group(2) == "393216":
            v[m.group(3)] = float(m.group(4))
        if "igemm3x" in m.group(1) and m.group(2) == "409600":
            v3[m.group(3)] = float(m.group(4))
    if v:
        v["_rows"] = 48768
        v["_kernel"] = "igemm4x_kernel (multi-stage LDS-DMA, 256 x 128 tiles; the three whole rounds = 48 768 of the 50 400 rows; the other rows: split-K launch of igemm3x)"
        return v
    v3["_rows"] = 50400
    v3["_kernel"] = "igemm3x_kernel<2,2,AP,OP>"
    return v3


def main():
    out = sys.argv[1]
    res = {}
    for arg in sys.argv[2:]:
        key, path = arg.split("=", 1)
        if key == "f32-from":
            res["f32"] = json.load(open(path))["f32"]
            continue
        v = parse(path)
        rows, cin, cout = int(v["_rows"]), 512, 512
        algo = rows * cin * 4 + rows * cout * 4 + 9 * cin * cout * 4  # gathered tensor once + output once + weight planes once
        res[key] = {
            "hbm_bytes_per_launch": int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024),
            "fetch_bytes": int(v["FETCH_SIZE"] * 1024 * 2), "write_bytes": int(v["WRITE_SIZE"] * 1024),
            "algorithmic_bytes_per_launch": algo,
            "tcc_hit_rate": v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]),
            "launch": v["_kernel"] + " fwd / bwd-data on packed planes, shared regression-head conv 3x3 512->512 over P3|P4|P5 at "
                      "batch 8 (%d rows of this launch); plane format %s" % (rows, "P16 (f16c8: the format this launch has in the "
                      "default mixed step)" if key != "bf16x3" else "bf16 pairs (bf16x3)"),
            "source": "%s (tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in separate passes on "
                      "tools/conv_bench.py --shape reg,cls --mode fwd3pp,dgrad3pp --fmt %d)" % (path, 0 if key == "bf16x3" else 1),
            "note": "FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md; Infinity-Cache hits are included in the counter, so this is an upper bound "
                    "on true HBM bytes.  Several times the algorithmic bytes: every workgroup streams the 2.4 MB of weight planes of its 128 "
                    "output channels and the workgroups of an XCD drift apart in phase, so the 9.4 MB of weights are re-fetched by every "
                    "round of workgroups on every XCD; at ~0.45-0.55 ms per launch this stays under a quarter of the HBM peak -- the launch "
                    "is bound by the matrix pipe and its feed, not by HBM.",
            "kernel_source_sha16": sha(["conv3.hip", "conv3_shared.h", "planes_fmt.h", "p16.h"]),
        }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: (v["hbm_bytes_per_launch"], v.get("tcc_hit_rate")) for k, v in res.items()}))


if __name__ == "__main__":
    main()
