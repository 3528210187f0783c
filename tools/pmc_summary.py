#!/usr/bin/env python3
"""Summarize rocprofv3 --pmc passes (separate runs of the same command) into one per-kernel table.

    python tools/pmc_summary.py <counter_collection.csv> [<counter_collection.csv> ...] > profiles/<tag>_pmc.csv

Every counter found is averaged per kernel launch.  When FETCH_SIZE and WRITE_SIZE are both present the HBM traffic
column is   (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes   (gfx950: FETCH_SIZE reports half the bytes of wide coalesced
reads, MI355X_MICROARCH.md section HBM; both counters are in KiB).  `mfma_busy_frac` = SQ_VALU_MFMA_BUSY_CYCLES /
SQ_BUSY_CYCLES when both are present (share of the shader-busy cycles in which a matrix instruction was executing)."""
import csv
import re
import sys
from collections import defaultdict


def kname(s):
    s = s.split("(")[0]
    s = re.sub(r"^void\s+", "", s)
    return s.replace("tgsr::", "")


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))     # kernel -> counter -> [n, sum]
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            a = acc[kname(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    counters = sorted({c for k in acc for c in acc[k]})
    cols = ["kernel", "launches"] + ["avg_" + c for c in counters]
    hbm = "FETCH_SIZE" in counters and "WRITE_SIZE" in counters
    busy = "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "SQ_BUSY_CYCLES" in counters
    if hbm:
        cols.append("avg_HBM_MB_per_launch")
    if busy:
        cols.append("mfma_busy_frac")
    print(",".join(cols))

    def total(k):
        return sum(v[1] for v in acc[k].values())

    for k in sorted(acc, key=lambda k: -acc[k].get("FETCH_SIZE", [0, total(k)])[1]):
        n = max(v[0] for v in acc[k].values())
        avg = {c: (acc[k][c][1] / acc[k][c][0] if c in acc[k] else float("nan")) for c in counters}
        row = ['"%s"' % k, str(n)] + ["%.1f" % avg[c] for c in counters]
        if hbm:
            row.append("%.3f" % ((2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024 / 1e6))
        if busy:
            row.append("%.4f" % (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / max(avg["SQ_BUSY_CYCLES"], 1.0)))
        print(",".join(row))


if __name__ == "__main__":
    main()
