#!/bin/bash
# Kernel timeline of hipGraph replays of the reduced-precision step: rocprofv3 kernel trace -> per-step busy time, gaps.
#   bash tools/graph_timeline.sh [bench args]   (on the GPU box; writes gpurun_out/graph_timeline.txt)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
RAW=/tmp/prof_tl
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d $RAW -o tl -- python3 $ROOT/bench.py --dtype bf16 --graph --steps 8 --warmup 2 --no-cpu-baseline --profile-every 0 "$@" > $OUT/graph_timeline.log 2>&1
KT=$(find $RAW -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/graph_timeline.py $KT > $OUT/graph_timeline.txt
rm -rf $RAW
