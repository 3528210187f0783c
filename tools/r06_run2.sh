#!/bin/bash
# round 6: kernel timelines of the train step, replayed from hipGraphs and eager (G-only and G/D)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in g1 g0 gan1 gan0; do
  case $mode in g1) export TGSR_GRAPH_G=1; extra="";; g0) export TGSR_GRAPH_G=0; extra="";; gan1) export TGSR_GRAPH_G=1; extra="--gan";; gan0) export TGSR_GRAPH_G=0; extra="--gan";; esac
  RAW=/tmp/prof_$mode; rm -rf $RAW; mkdir -p $RAW
  timeout -k 10 400 rocprofv3 --output-format csv --kernel-trace -d $RAW -o tl -- python3 $ROOT/bench.py --mode train $extra --steps 6 --warmup 4 --repeats 1 --no-cpu-baseline > $OUT/tl_$mode.log 2>&1 || { echo "rocprof $mode failed"; tail -5 $OUT/tl_$mode.log; exit 1; }
  KT=$(find $RAW -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/tools/train_timeline.py $KT > $OUT/timeline_$mode.txt
  head -4 $OUT/timeline_$mode.txt
  rm -rf $RAW
done
cd $ROOT && timeout -k 10 600 python -m pytest tests/test_hip_gan.py tests/test_hip_dp.py -x -q -k "graph_replayed or gloo" > $OUT/t2.log 2>&1; echo "pytest rc=$?"; tail -n 15 $OUT/t2.log
