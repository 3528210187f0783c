#!/bin/bash
# 1x1 convolutions of CNN_ENCODER's heads on the implicit-GEMM kernel: parity, then the C3 and DAMSM pre-training lines with and without.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_train.py tests/test_hip_custom_ops.py tests/test_hip_gan.py tests/test_hip_dp.py -x -q -k "conv1x1 or heads or cnn_encoder or implicit_gemm or linear or lstm or bilstm or gru or ca_net or attention or damsm or DAMSM or opcheck or encoder or dp or two_rank" > $OUT/t41.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 6 $OUT/t41.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for c in 1 0 1 0; do
  TGSR_CONV1X1_GCONV=$c timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/c1x1_${c}.json 2> /dev/null; echo "enc c=$c rc=$?"
  TGSR_CONV1X1_GCONV=$c timeout -k 10 300 python bench.py --mode damsm --steps 20 > $OUT/c1x1d_${c}.json 2> /dev/null; echo "damsm c=$c rc=$?"
  python - $OUT/c1x1_${c}.json $OUT/c1x1d_${c}.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"])
PY
done
