#!/bin/bash
# Trunk branch streams: parity (bit-identity with the one-stream walk, eager and captured), the measured graph policy, the C3 line
# with 1 and 4 trunk streams; aten launches of a G/D step.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_inception.py tests/test_hip_train.py tests/test_hip_gan.py -x -q -k "trunk or streams or pool or graph or policy" > $OUT/t11.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 12 $OUT/t11.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for n in 1 4; do
  TGSR_TRUNK_STREAMS=$n timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/train_enc_streams$n.json 2> $OUT/train_enc_streams$n.err; echo "streams=$n rc=$?"
  python - "$OUT/train_enc_streams$n.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("final_loss"), d.get("graph_policy"))
except Exception as e: print("no line", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
timeout -k 10 300 python bench.py --mode train --steps 10 --no-cpu-baseline > $OUT/train_g_auto.json 2> $OUT/train_g_auto.err; echo "g rc=$?"
timeout -k 10 300 python bench.py --mode train --gan --steps 10 --no-cpu-baseline > $OUT/train_gan_auto.json 2> $OUT/train_gan_auto.err; echo "gan rc=$?"
python - $OUT/train_g_auto.json $OUT/train_gan_auto.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d.get("graph_policy"), d.get("launch"))
    except Exception as e: print("no line", f, e)
PY
TGSR_GRAPH_G=0 TGSR_GRAPH_D=0 timeout -k 10 300 python tools/gan_host_ops.py > $OUT/gan_host_ops.txt 2> $OUT/gan_host_ops.err; echo "host ops rc=$?"
head -45 $OUT/gan_host_ops.txt | cut -c1-250
