"""Summary of one training step from a rocprofv3 --kernel-trace CSV of bench.py (steps are delimited by adam_kernel):
wall time, time with 0 / 1 / 2 kernels running, per-queue busy time in the forward and backward phases, kernel time by name.
usage: python tools/trace_summary.py <..._kernel_trace.csv> [step index]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) - 2
seg = rows[adam[k] + 1: adam[k + 1] + 1]
t0, t1 = rows[adam[k]]["e"], seg[-1]["e"]


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "")[:64]


def busy(rs):
    ev = sorted((r["s"], r["e"]) for r in rs)
    b, (cs, ce) = 0, ev[0]
    for s, e in ev[1:]:
        if s > ce:
            b, cs, ce = b + ce - cs, s, e
        else:
            ce = max(ce, e)
    return b + ce - cs


print("step %d: %d launches, wall %.3f ms, some kernel running %.3f ms, sum of kernel durations %.3f ms"
      % (k, len(seg), (t1 - t0) / 1e6, busy(seg) / 1e6, sum(r["e"] - r["s"] for r in seg) / 1e6))
pts = sorted([(r["s"], 1) for r in seg] + [(r["e"], -1) for r in seg])
lvl, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[lvl] += t - last
    last, lvl = t, lvl + d
print("time with n kernels running (ms):", {n: round(v / 1e6, 3) for n, v in sorted(hist.items())})
first_w = min((r["s"] for r in seg if "wgrad" in r["Kernel_Name"]), default=t1)
for name, lo, hi in (("forward + losses", t0, first_w), ("backward + optimizer", first_w, t1)):
    for q in sorted({r["Queue_Id"] for r in seg}):
        rs = [r for r in seg if r["Queue_Id"] == q and lo <= r["s"] < hi]
        if rs:
            print("%-22s queue %s: %3d launches, busy %.2f ms of %.2f" % (name, q, len(rs), busy(rs) / 1e6, (hi - lo) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    a = agg[short(r["Kernel_Name"])]
    a[0] += 1
    a[1] += r["e"] - r["s"]
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%4d  %9.1f us  %s" % (c, t / 1e3, n))
