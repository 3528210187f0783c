#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
for fill in 320 224 448 320 224; do
  TGSR_GCONV_FILL=$fill timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/fillb_$fill.json 2> /dev/null; echo "fill=$fill rc=$?"
  python - "$OUT/fillb_$fill.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])
except Exception as e: print("no line", e)
PY
done
