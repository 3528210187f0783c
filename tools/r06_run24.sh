#!/bin/bash
# Timelines: a replayed generator train step; the fp32 one-lane inference replay.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
TL_WHICH=--middle bash tools/train_timeline.sh > $OUT/r06_tl3.log 2>&1; echo "timeline g rc=$?"
cp $OUT/train_timeline.txt $OUT/r06_train_timeline_g_replay.txt
head -45 $OUT/r06_train_timeline_g_replay.txt | cut -c1-160
