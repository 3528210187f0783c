#!/bin/bash
# DAMSM backward with the attention chunks staged in LDS: parity (the damsm goldens), its time, the C3 line.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_train.py tests/test_hip_custom_ops.py -x -q -k "damsm or words or sent or opcheck" > $OUT/t19.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 6 $OUT/t19.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
python tools/bench_damsm_bwd.py
timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/enc_damsm_lds.json 2> $OUT/enc_damsm_lds.err; echo "enc rc=$?"
python - $OUT/enc_damsm_lds.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"], d.get("graph_policy"))
PY
