#!/bin/bash
# Kernel timeline of the training step: bash tools/train_timeline.sh [bench args]   (GPU box; writes gpurun_out/train_timeline.txt)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
RAW=/tmp/prof_ttl
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d $RAW -o tl -- python3 $ROOT/bench.py --mode train --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline "$@" > $OUT/train_timeline.log 2>&1
KT=$(find $RAW -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/train_timeline.py $KT $TL_WHICH > $OUT/train_timeline.txt
rm -rf $RAW
