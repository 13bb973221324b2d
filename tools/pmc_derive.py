#!/usr/bin/env python3
"""Derived figures from a tools/pmc_summary.py table (the output of tools/pmc_stalls.sh / tools/pmc_backbone.sh): per kernel and grid
   MFMA-busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)   (GUI_ACTIVE is summed over the 8 XCDs)
   and, per wave-cycle (SQ_WAVE_CYCLES): waiting on dependencies (SQ_WAIT_INST_ANY), of which LDS (SQ_WAIT_INST_LDS), parked in
   s_waitcnt / barriers (SQ_WAIT_ANY), issuing (SQ_ACTIVE_INST_ANY); LDS index unit active per CU-busy cycle; VALU per MFMA.
usage: python3 tools/pmc_derive.py gpurun_out/stalls_TAG.txt [...]"""
import collections
import re
import sys


def main():
    tab = collections.OrderedDict()
    for path in sys.argv[1:]:
        for line in open(path):
            m = re.match(r"(\S.*?)\s+grid=(\d+)\s+(\S+)\s+n=(\d+) mean=(\S+)", line)
            if m:
                tab.setdefault((m.group(1).strip(), int(m.group(2))), {})[m.group(3)] = float(m.group(5))
    print("%-52s %9s | %9s %8s %8s %8s %8s | %8s %9s" % ("kernel (grid = threads)", "MFMA-busy", "wait-dep", "of: LDS", "parked", "issuing", "LDS-unit", "VALU/MFMA", "MFMA inst"))
    for (k, g), c in tab.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or c.get("SQ_INSTS_MFMA", 1) == 0:
            continue
        busy = "%8.1f%%" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)) if "GRBM_GUI_ACTIVE" in c else "      n/a"
        wc = c.get("SQ_WAVE_CYCLES")

        def per(n):
            return "%8.2f" % (c[n] / wc) if (wc and n in c) else "     n/a"
        lds = "%8.2f" % (c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"]) if ("SQ_LDS_IDX_ACTIVE" in c and "SQ_BUSY_CU_CYCLES" in c) else "     n/a"
        vm = "%8.1f" % (c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"] - 1) if ("SQ_INSTS_VALU" in c and c.get("SQ_INSTS_MFMA")) else "     n/a"
        print("%-52s %9s | %9s %8s %8s %8s %8s | %8s %9.3g" % (("%s grid=%d" % (k, g))[:52], busy, per("SQ_WAIT_INST_ANY"), per("SQ_WAIT_INST_LDS"),
                                                             per("SQ_WAIT_ANY"), per("SQ_ACTIVE_INST_ANY"), lds, vm, c.get("SQ_INSTS_MFMA", 0)))


if __name__ == "__main__":
    main()
