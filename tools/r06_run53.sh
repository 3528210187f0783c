#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_inception.py tests/test_hip_train.py tests/test_hip_parity.py tests/test_hip_gan.py tests/test_hip_custom_ops.py -x -q -k "inception or gconv or trunk or streams or stride2 or conv1x1 or heads or cnn_encoder or damsm or DAMSM or opcheck or encoder or full_size_gan or linear or implicit" > $OUT/t53.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 5 $OUT/t53.log | cut -c1-220
exit $rc
