#!/bin/bash
# rocprofv3 passes over an arbitrary python command (run on the GPU box through gpurun).  Usage:
#   bash tools/profile_pmc_cmd.sh <tag> <script.py> [args...]
# Leaves gpurun_out/<tag>_kernel_stats.csv and gpurun_out/<tag>_pmc.csv (per-kernel averages of every counter).
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $RAW/stats -o $TAG -- python3 $SCRIPT "$@" > $OUT/${TAG}_stats.log 2>&1
KS=$(find $RAW/stats -name "*kernel_stats*.csv" | head -1)
python3 $ROOT/tools/trim_stats.py $KS > $OUT/${TAG}_kernel_stats.csv
echo "[$TAG] stats done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $RAW/fetch -o $TAG -- python3 $SCRIPT "$@" > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $RAW/write -o $TAG -- python3 $SCRIPT "$@" > $OUT/${TAG}_write.log 2>&1
echo "[$TAG] traffic done"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $RAW/mfma -o $TAG -- python3 $SCRIPT "$@" > $OUT/${TAG}_mfma.log 2>&1 || echo "[$TAG] mfma counter pass failed (see log)"
echo "[$TAG] mfma done"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $RAW/sq -o $TAG -- python3 $SCRIPT "$@" > $OUT/${TAG}_sq.log 2>&1 || echo "[$TAG] sq counter pass failed (see log)"
echo "[$TAG] sq done"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES -d $RAW/sq2 -o $TAG -- python3 $SCRIPT "$@" > $OUT/${TAG}_sq2.log 2>&1 || echo "[$TAG] sq2 counter pass failed (see log)"
echo "[$TAG] sq2 done"
python3 $ROOT/tools/pmc_summary.py --stats $KS $(find $RAW/fetch $RAW/write $RAW/mfma $RAW/sq $RAW/sq2 -name "*counter_collection.csv") > $OUT/${TAG}_pmc.csv
rm -rf $RAW
