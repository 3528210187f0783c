#!/usr/bin/env python3
"""Summarize rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of the same command) into per-kernel HBM
traffic:  python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv>
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads,
MI355X_MICROARCH.md section HBM)."""
import csv, sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"].split("(")[0]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    print("kernel,launches,avg_FETCH_SIZE_KB,avg_WRITE_SIZE_KB,avg_HBM_MB_per_launch")
    for k in sorted(f, key=lambda k: -f[k][1]):
        n = f[k][0]
        fk = f[k][1] / n
        wk = w[k][1] / max(1, w[k][0]) if k in w else 0.0
        print('"%s",%d,%.1f,%.1f,%.2f' % (k, n, fk, wk, (2 * fk + wk) * 1024 / 1e6))


if __name__ == "__main__":
    main()
