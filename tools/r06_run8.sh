#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/prof_enc; rm -rf $RAW; mkdir -p $RAW
timeout -k 10 500 rocprofv3 --output-format csv --kernel-trace --stats -d $RAW -o enc -- python3 $ROOT/bench.py --mode train --gan --damsm-encoder --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline > $OUT/prof_enc.log 2>&1 || { echo rocprof failed; tail -5 $OUT/prof_enc.log; }
ST=$(find $RAW -name "*kernel_stats.csv" | head -1)
python3 $ROOT/tools/trim_stats.py $ST > $OUT/enc6_kernel_stats.csv
python3 - $OUT/enc6_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    print("%-80s %5s %8.1f us avg %8.2f ms %5.1f%%"%(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
