#!/bin/bash
# DAMSM pair backward with 512 threads per pair: the damsm goldens / train / dp tests, its time, the pre-training and C3 lines.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_train.py tests/test_hip_custom_ops.py tests/test_hip_dp.py -x -q -k "damsm or words or sent or opcheck or DAMSM or dp or two_rank" > $OUT/t36.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 6 $OUT/t36.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
python tools/bench_damsm_bwd.py
timeout -k 10 300 python bench.py --mode damsm --steps 20 > $OUT/damsm512.json 2> $OUT/damsm512.err; echo "damsm rc=$?"
timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/enc_damsm512.json 2> $OUT/enc_damsm512.err; echo "enc rc=$?"
python - $OUT/damsm512.json $OUT/enc_damsm512.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d.get("final_loss"))
PY
