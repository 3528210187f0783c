#!/bin/bash
# Timeline of the C3 step (who runs alone), and the split target above the default.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
for fill in 480 640; do
  TGSR_GCONV_FILL=$fill timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/train_enc_fill$fill.json 2> $OUT/train_enc_fill$fill.err; echo "fill=$fill rc=$?"
  python - "$OUT/train_enc_fill$fill.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("graph_policy"))
except Exception as e: print("no line", e)
PY
done
bash tools/train_timeline.sh --gan --damsm-encoder > $OUT/tl.log 2>&1; echo "timeline rc=$?"
cp $ROOT/gpurun_out/train_timeline.txt $OUT/train_timeline_enc.txt
head -70 $OUT/train_timeline_enc.txt | cut -c1-200
