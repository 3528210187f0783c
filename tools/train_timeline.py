#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py --mode train: per step (delimited by the text encoder's recurrence) the wall time, the
device-busy union, the time only ONE kernel is in flight, idle time, kernel time per queue, and the kernels that run alone the
longest - those are what a shorter step has to come from."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
q = "Queue_Id" if "Queue_Id" in rows[0] else None
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(q, "0") if q else "0") for r in rows), key=lambda e: e[0])
marks = [i for i, e in enumerate(ev) if "lstm_recurrent" in e[2]]
if len(marks) < 3:
    sys.exit("not enough steps in the trace")
def short(n):
    n = n.replace("void ", "").replace("tgsr::", "")
    return n[:70]
# which steps: the last two by default; `--middle` (argv[2]) = two steps from the middle of the run (bench.py's timed region - the last
# steps of a bench run are its event-sampled eager ones)
if len(sys.argv) > 2 and sys.argv[2] == "--middle":
    mid = len(marks) // 2
    marks = marks[:mid + 2]
for a, b in zip(marks[-3:-1], marks[-2:]):
    step = ev[a:b]
    t0, t1 = step[0][0], max(e[1] for e in step)
    pts = sorted([(s, 1, i) for i, (s, e, _, _) in enumerate(step)] + [(e, -1, i) for i, (s, e, _, _) in enumerate(step)])
    busy = single = 0
    live = set()
    last = pts[0][0]
    alone = collections.Counter()
    for t, d, i in pts:
        if len(live) >= 1:
            busy += t - last
        if len(live) == 1:
            single += t - last
            alone[short(step[next(iter(live))][2])] += t - last
        if d > 0:
            live.add(i)
        else:
            live.discard(i)
        last = t
    per_q = collections.Counter()
    for s, e, n, qq in step:
        per_q[qq] += e - s
    print("step: %d kernels, %.2f ms to the next step; kernel time %.2f ms; busy %.2f, exactly one kernel in flight %.2f, idle %.2f ms"
          % (len(step), (ev[b][0] - t0) / 1e6, sum(e - s for s, e, _, _ in step) / 1e6, busy / 1e6, single / 1e6, (ev[b][0] - t0 - busy) / 1e6))
    print("  kernel time per queue (ms):", {k: round(v / 1e6, 2) for k, v in per_q.items()})
print("\nlast step: time spent ALONE on the device, by kernel (ms):")
for n, v in alone.most_common(25):
    print("  %7.3f  %s" % (v / 1e6, n))
# gaps: idle intervals > 3 us and what follows them
step = ev[marks[-2]:marks[-1]]
latest = step[0][0]
gaps = []
for s, e, n, qq in step:
    if s - latest > 3000:
        gaps.append((s - latest, short(n)))
    latest = max(latest, e)
print("\nidle gaps > 3 us in the last step: %d, %.3f ms in all; the largest:" % (len(gaps), sum(g for g, _ in gaps) / 1e6))
for g, n in sorted(gaps, reverse=True)[:15]:
    print("  %6.1f us before %s" % (g / 1e3, n))
