#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_inception.py -x -q > $OUT/t5.log 2>&1
echo "pytest rc=$?"; tail -n 40 $OUT/t5.log
timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/train_enc_hip.json 2> $OUT/train_enc_hip.err; echo "rc=$?"
TGSR_TRUNK=torch timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/train_enc_torch.json 2> $OUT/train_enc_torch.err; echo "rc=$?"
for f in train_enc_hip train_enc_torch; do python - "$OUT/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d["final_loss"])
except Exception as e: print("no line", e); print(open(sys.argv[1].replace(".json",".err")).read()[-2000:])
PY
done
