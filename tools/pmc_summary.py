#!/usr/bin/env python3
"""Summarize rocprofv3 --pmc passes (separate runs of the same command) into one per-kernel table.

    python tools/pmc_summary.py <counter_collection.csv> [<counter_collection.csv> ...] > profiles/<tag>_pmc.csv

Every counter found is averaged per kernel launch.  When FETCH_SIZE and WRITE_SIZE are both present the HBM traffic
column is   (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes   (gfx950: FETCH_SIZE reports half the bytes of wide coalesced
reads, MI355X_MICROARCH.md section HBM; both counters are in KiB).  With `--stats <kernel_stats.csv>` (the rocprofv3
--stats summary of the SAME command) two more columns: `avg_us` and `mfma_busy_frac_of_peak` =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x avg duration x 2.4 GHz): the share of the chip's matrix-pipe cycles at the spec
clock in which an MFMA was executing (SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the SIMDs)."""
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
    args, dur = sys.argv[1:], {}
    if "--stats" in args:
        i = args.index("--stats")
        for r in csv.DictReader(open(args[i + 1])):
            dur[kname(r["Name"])] = float(r["AverageNs"])
        del args[i:i + 2]
    for path in args:
        for r in csv.DictReader(open(path)):
            a = acc[kname(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    counters = sorted({c for k in acc for c in acc[k]})
    cols = ["kernel", "launches"] + ["avg_" + c for c in counters]
    hbm = "FETCH_SIZE" in counters and "WRITE_SIZE" in counters
    busy = "SQ_VALU_MFMA_BUSY_CYCLES" in counters and bool(dur)
    if hbm:
        cols.append("avg_HBM_MB_per_launch")
    if busy:
        cols += ["avg_us", "mfma_busy_frac_of_peak"]
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
            ns = dur.get(k, float("nan"))
            row += ["%.2f" % (ns / 1e3), "%.4f" % (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * ns * 2.4))]
        print(",".join(row))


if __name__ == "__main__":
    main()
