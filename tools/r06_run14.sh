#!/bin/bash
# Stride-2 data gradients by parity classes: parity, then the C3 line with and without.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_inception.py tests/test_hip_custom_ops.py -x -q > $OUT/t14.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 12 $OUT/t14.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for c in 1 0; do
  TGSR_TRUNK_CLASS_DGRAD=$c timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/train_enc_cls$c.json 2> $OUT/train_enc_cls$c.err; echo "class=$c rc=$?"
  python - "$OUT/train_enc_cls$c.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("final_loss"), d.get("graph_policy"))
except Exception as e: print("no line", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
