#!/bin/bash
# Timeline of a REPLAYED C3 step (the middle of the timed region) and of the G/D step.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
TL_WHICH=--middle bash tools/train_timeline.sh --gan --damsm-encoder > $OUT/tl.log 2>&1; echo "timeline rc=$?"
cp $ROOT/gpurun_out/train_timeline.txt $OUT/train_timeline_enc_mid.txt
head -64 $OUT/train_timeline_enc_mid.txt | cut -c1-200
