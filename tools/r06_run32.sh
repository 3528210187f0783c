#!/bin/bash
# Trunk backward: the branch heads' contributions in slots on their own streams (sum behind the join) against heads behind the join.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_inception.py tests/test_hip_custom_ops.py -x -q > $OUT/t32.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 8 $OUT/t32.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_hip_gan.py -x -q -k "full_size_gan_train_step_parity" > $OUT/t32b.log 2>&1; echo "gan parity rc=$?"; tail -n 3 $OUT/t32b.log | cut -c1-200
for c in 1 0 1 0; do
  TGSR_TRUNK_HEADS_PARALLEL=$c timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/heads${c}.json 2> $OUT/heads${c}.err; echo "heads=$c rc=$?"
  python - "$OUT/heads${c}.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("graph_policy"))
except Exception as e: print("no line", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
