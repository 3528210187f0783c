#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
TL_WHICH=--middle bash tools/train_timeline.sh --gan --damsm-encoder > $OUT/r06_tl.log 2>&1; echo "timeline rc=$?"
cp $OUT/train_timeline.txt $OUT/r06_train_timeline_enc_replay.txt
sed -n 1,45p $OUT/r06_train_timeline_enc_replay.txt | cut -c1-150
