#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_optim.py tests/test_hip_train.py tests/test_hip_dp.py tests/test_hip_gan.py -x -q -k "adam or graph or policy or dp or two_rank or step" > $OUT/t40.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 6 $OUT/t40.log | cut -c1-220
exit $rc
