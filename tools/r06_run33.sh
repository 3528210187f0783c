#!/bin/bash
# The image encoder issued before the discriminator updates, on a stream / as a graph of its own: parity, then the C3 line with and without.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_gan.py -x -q -k "image_encoder_beside or full_size_gan_train_step_parity or graph_replayed" > $OUT/t33.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 10 $OUT/t33.log | cut -c1-250
[ $rc -eq 0 ] || exit $rc
for c in 1 0 1 0; do
  TGSR_ENC_EARLY=$c timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/early${c}.json 2> $OUT/early${c}.err; echo "early=$c rc=$?"
  python - "$OUT/early${c}.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("graph_policy"))
except Exception as e: print("no line", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
