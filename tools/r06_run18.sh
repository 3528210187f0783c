#!/bin/bash
# Discriminator weight gradients on side streams: the gan / dp suites, then the G/D and C3 lines with and without.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests/test_hip_gan.py tests/test_hip_dp.py tests/test_hip_train.py -x -q > $OUT/t18.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 8 $OUT/t18.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for c in 1 0; do
  TGSR_D_WGRAD_SIDE=$c timeout -k 10 400 python bench.py --mode train --gan --steps 10 --no-cpu-baseline > $OUT/dws${c}_gan.json 2> $OUT/dws${c}_gan.err; echo "side=$c gan rc=$?"
  TGSR_D_WGRAD_SIDE=$c timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/dws${c}_enc.json 2> $OUT/dws${c}_enc.err; echo "side=$c enc rc=$?"
done
python - $OUT <<'PY'
import json,sys,os
for c in (1,0):
    for n in ("gan","enc"):
        f=os.path.join(sys.argv[1],"dws%d_%s.json"%(c,n))
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); print(c, n, d["ms_per_step"], d["value"], d.get("final_loss"), (d.get("graph_policy") or {}))
        except Exception as e: print("no line", f, e)
PY
