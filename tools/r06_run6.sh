#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_hip_inception.py -x -q > $OUT/t6.log 2>&1
echo "pytest rc=$?"; tail -n 12 $OUT/t6.log
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/prof_enc; rm -rf $RAW; mkdir -p $RAW
timeout -k 10 500 rocprofv3 --output-format csv --kernel-trace --stats -d $RAW -o enc -- python3 $ROOT/bench.py --mode train --gan --damsm-encoder --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline > $OUT/prof_enc.log 2>&1 || { echo rocprof failed; tail -5 $OUT/prof_enc.log; }
ST=$(find $RAW -name "*kernel_stats.csv" | head -1)
python3 $ROOT/tools/trim_stats.py $ST > $OUT/enc_kernel_stats.csv 2>/dev/null || cp $ST $OUT/enc_kernel_stats.csv
head -40 $OUT/enc_kernel_stats.csv | cut -c1-200
