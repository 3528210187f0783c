#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_dp.py tests/test_hip_train.py -x -q > $OUT/t25.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 8 $OUT/t25.log | cut -c1-220
exit $rc
