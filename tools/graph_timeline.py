#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py --graph: the last replays' timeline - wall per step, union of busy time,
time with >= 2 kernels in flight, idle time, and the gaps in front of each kernel on the longest chain."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# steps are delimited by the text encoder's recurrent kernel (once per step)
marks = [i for i, e in enumerate(ev) if "lstm_recurrent" in e[2]]
if len(marks) < 4:
    sys.exit("not enough steps in the trace")
for a, b in zip(marks[-4:-1], marks[-3:]):
    step = ev[a:b]
    t0, t1 = step[0][0], max(e[1] for e in step)
    # sweep
    pts = sorted([(s, 1) for s, e, _ in step] + [(e, -1) for s, e, _ in step])
    busy = multi = 0
    depth, last = 0, pts[0][0]
    for t, d in pts:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    ksum = sum(e - s for s, e, _ in step)
    print("step: %d kernels, first start -> last end %.1f us, to the next step's start %.1f us; kernel time sum %.1f us; device busy "
          "(>= 1 kernel) %.1f us, >= 2 kernels %.1f us, idle inside the step %.1f us"
          % (len(step), (t1 - t0) / 1e3, (ev[b][0] - t0) / 1e3, ksum / 1e3, busy / 1e3, multi / 1e3, (t1 - t0 - busy) / 1e3))
step = ev[marks[-2]:marks[-1]]
t0 = step[0][0]
print("\nlast full step, in start order (start us, duration us, gap since the latest earlier end us):")
latest_end = step[0][0]
for s, e, n in step:
    print("%8.1f %7.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - latest_end) / 1e3, n[:90]))
    latest_end = max(latest_end, e)
