#!/bin/bash
# The trunk's split target beside branch streams (TGSR_GCONV_FILL), then the kernel statistics of the C3 step.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
for fill in 320 160 96; do
  TGSR_GCONV_FILL=$fill timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/train_enc_fill$fill.json 2> $OUT/train_enc_fill$fill.err; echo "fill=$fill rc=$?"
  python - "$OUT/train_enc_fill$fill.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("graph_policy"))
except Exception as e: print("no line", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/prof_enc; rm -rf $RAW; mkdir -p $RAW
TGSR_GRAPH_G=0 timeout -k 10 500 rocprofv3 --output-format csv --kernel-trace --stats -d $RAW -o enc -- python3 $ROOT/bench.py --mode train --gan --damsm-encoder --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline > $OUT/prof_enc.log 2>&1 || { echo rocprof failed; tail -5 $OUT/prof_enc.log; }
ST=$(find $RAW -name "*kernel_stats.csv" | head -1)
python3 $ROOT/tools/trim_stats.py $ST > $OUT/enc_kernel_stats.csv
python3 - $OUT/enc_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:32]:
    print("%-90s %5s %8.1f us avg %8.2f ms %5.1f%%"%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
