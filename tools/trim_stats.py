#!/usr/bin/env python3
"""Shorten the kernel names of a rocprofv3 kernel_stats CSV (torch's templated names run to kilobytes) so the summary
can be committed under profiles/:  python tools/trim_stats.py in.csv > out.csv"""
import csv
import sys

w = csv.writer(sys.stdout, quoting=csv.QUOTE_MINIMAL)
for i, r in enumerate(csv.reader(open(sys.argv[1]))):
    if i and len(r[0]) > 160:
        r[0] = r[0][:157] + "..."
    w.writerow(r)
