#!/bin/bash
# The longest individual non-tgsr (aten / runtime) kernels of a training step: bash tools/train_top_aten.sh [bench args]
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
RAW=/tmp/prof_tta
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d $RAW -o tl -- python3 $ROOT/bench.py --mode train --steps 8 --warmup 5 --repeats 1 --no-cpu-baseline "$@" > $OUT/train_top_aten.log 2>&1
KT=$(find $RAW -name "*kernel_trace.csv" | head -1)
python3 - "$KT" > $OUT/train_top_aten.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))) for r in rows]
n = len(ev)
ev = [e for e in ev if "tgsr::" not in e[1]]
agg = collections.defaultdict(lambda: [0, 0])
for d, k, g, w in ev:
    a = agg[(k[:110], g)]
    a[0] += 1
    a[1] += d
print("non-tgsr kernels by total time over the whole run (%d kernels in all): count, total us, avg us, grid, name" % n)
for (k, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%6d %10.1f %8.1f  grid %-10s %s" % (c, t / 1e3, t / 1e3 / c, g, k))
PY
rm -rf $RAW
