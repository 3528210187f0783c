#!/bin/bash
# rocprofv3 passes over one bench command (run on the GPU box through gpurun).  Usage:
#   bash tools/profile_pmc.sh <tag> <bench args...>
# Leaves in gpurun_out/: <tag>_kernel_stats.csv (rocprofv3 --output-format csv --kernel-trace --stats summary), <tag>_pmc.csv (per-kernel
# averages of every counter, tools/pmc_summary.py), <tag>_bench.json (the bench line of the stats run) and the logs.
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE
# 2 - they cannot share a pass; gpurun refuses --pmc together with the hip/hsa trace domains).  Raw traces are deleted
# (tens of MB): only the summaries travel back.
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
if [ -z "$KS" ]; then find $RAW/stats | head -30; else python3 $ROOT/tools/trim_stats.py $KS > $OUT/${TAG}_kernel_stats.csv; fi
echo "[$TAG] stats done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $RAW/fetch -o $TAG -- python3 $ROOT/bench.py "$@" > $OUT/${TAG}_fetch.log 2>&1
echo "[$TAG] fetch done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $RAW/write -o $TAG -- python3 $ROOT/bench.py "$@" > $OUT/${TAG}_write.log 2>&1
echo "[$TAG] write done"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $RAW/mfma -o $TAG -- python3 $ROOT/bench.py "$@" > $OUT/${TAG}_mfma.log 2>&1 || echo "[$TAG] mfma counter pass failed (see log)"
echo "[$TAG] mfma done"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $RAW/sq -o $TAG -- python3 $ROOT/bench.py "$@" > $OUT/${TAG}_sq.log 2>&1 || echo "[$TAG] sq counter pass failed (see log)"
echo "[$TAG] sq done"
python3 $ROOT/tools/pmc_summary.py --stats $KS $(find $RAW/fetch $RAW/write $RAW/mfma $RAW/sq -name "*counter_collection.csv") > $OUT/${TAG}_pmc.csv
head -12 $OUT/${TAG}_pmc.csv
rm -rf $RAW
