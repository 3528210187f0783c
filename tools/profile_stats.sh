#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench command, summary only:  bash tools/profile_stats.sh <tag> <bench args...>
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $RAW/stats -o $TAG -- python3 $ROOT/bench.py "$@" > $OUT/${TAG}_stats.log 2>&1
grep '^{' $OUT/${TAG}_stats.log | tail -1 > $OUT/${TAG}_bench.json || true
KS=$(find $RAW/stats -name "*kernel_stats*.csv" | head -1)
python3 $ROOT/tools/trim_stats.py $KS > $OUT/${TAG}_kernel_stats.csv
rm -rf $RAW
