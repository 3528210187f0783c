#!/bin/bash
# The image heads with their copies in flight: parity (bit-identity with the double-buffered kernel), then the headline with and without.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "conv_to3 or head or pipeline or golden or checkpoint" > $OUT/t26.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 8 $OUT/t26.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for p in 1 0 1 0; do
  TGSR_TO3_PIPE=$p timeout -k 10 300 python bench.py --no-cpu-baseline --extras none > $OUT/to3pipe${p}.json 2> $OUT/to3pipe${p}.err; echo "pipe=$p rc=$?"
  python - $OUT/to3pipe${p}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get("kernels") or {}
print(d["value"], d["ms_per_step"], d.get("value_throughput_form"), {n: v for n, v in k.items() if "to3" in n})
PY
done
