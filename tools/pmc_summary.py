"""Summarise rocprofv3 --pmc output directories: mean counter value per (kernel, counter) for our conv kernels.
usage: python tools/pmc_summary.py DIR [DIR ...]   (each DIR holds */*_counter_collection.csv)"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(igemm4[xp]_kernel|igemm3[fx]?_kernel|igemm2[fx]?_kernel|wgrad2f?_kernel|wgrad3[frw]?_kernel|igemm_kernel|wgrad_kernel|splitk_finish_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    acc = collections.defaultdict(list)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    acc[(k, r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (k, c, g), v in sorted(acc.items()):
        print("%-34s grid=%-8s %-28s n=%d mean=%.6g" % (k, g, c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
