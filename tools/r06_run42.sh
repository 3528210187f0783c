#!/bin/bash
# Kernel statistics of the DAMSM pre-training step.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/prof_damsm; rm -rf $RAW; mkdir -p $RAW
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $RAW -o d -- python3 $ROOT/bench.py --mode damsm --steps 20 > $OUT/r06_prof_damsm.log 2>&1 || { echo rocprof failed; tail -5 $OUT/r06_prof_damsm.log; }
ST=$(find $RAW -name "*kernel_stats.csv" | head -1)
python3 $ROOT/tools/trim_stats.py $ST > $OUT/r06_damsm_kernel_stats.csv
python3 - $OUT/r06_damsm_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:30]:
    print("%-90s %5s %8.1f us avg %8.2f ms %5.1f%%"%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
grep '^{' $OUT/r06_prof_damsm.log | tail -1 | cut -c1-200
