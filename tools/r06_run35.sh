#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 300 python bench.py --mode damsm --steps 20 > $OUT/r06_bench_damsm.json 2> $OUT/r06_bench_damsm.err; echo "damsm rc=$?"; tail -c 1200 $OUT/r06_bench_damsm.json
bash tools/r06_run22.sh
