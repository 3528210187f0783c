#!/bin/bash
# Round-6 profile set: rocprofv3 kernel statistics + PMC passes of the train, G/D, fp32 and bf16 lines (tools/final_profiles.sh prof),
# the kernel statistics of the G/D + DAMSM step and the timeline of a replayed one.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
bash tools/final_profiles.sh r06 prof > $OUT/r06_final_prof.log 2>&1; echo "prof rc=$?"; tail -5 $OUT/r06_final_prof.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/prof_enc; rm -rf $RAW; mkdir -p $RAW
timeout -k 10 500 rocprofv3 --output-format csv --kernel-trace --stats -d $RAW -o enc -- python3 $ROOT/bench.py --mode train --gan --damsm-encoder --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline > $OUT/r06_prof_enc.log 2>&1 || { echo rocprof failed; tail -5 $OUT/r06_prof_enc.log; }
ST=$(find $RAW -name "*kernel_stats.csv" | head -1)
python3 $ROOT/tools/trim_stats.py $ST > $OUT/r06_train_gan_damsm_kernel_stats.csv; echo "enc stats rows: $(wc -l < $OUT/r06_train_gan_damsm_kernel_stats.csv)"
rm -rf $RAW
cd $ROOT
TL_WHICH=--middle bash tools/train_timeline.sh --gan --damsm-encoder > $OUT/r06_tl.log 2>&1; echo "timeline rc=$?"
cp $OUT/train_timeline.txt $OUT/r06_train_timeline_enc_replay.txt
head -4 $OUT/r06_train_timeline_enc_replay.txt | cut -c1-200
TL_WHICH=--middle bash tools/train_timeline.sh --gan > $OUT/r06_tl2.log 2>&1; echo "timeline gan rc=$?"
cp $OUT/train_timeline.txt $OUT/r06_train_timeline_gan_replay.txt
head -4 $OUT/r06_train_timeline_gan_replay.txt | cut -c1-200
