#!/bin/bash
# round 6, first GPU pass: the graph-replayed generator update (tests) and the two train benches
set -o pipefail
mkdir -p gpurun_out/r06
timeout -k 10 900 python -m pytest tests/test_hip_train.py tests/test_hip_gan.py tests/test_hip_dp.py -x -q -k "graph_replayed or pack_cache or decreases_loss or gloo" > gpurun_out/r06/t1.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r06/t1.log
tail -n 30 gpurun_out/r06/t1.log
timeout -k 10 300 python bench.py --mode train --steps 10 --no-cpu-baseline > gpurun_out/r06/train_g.json 2> gpurun_out/r06/train_g.err; echo "train rc=$?"
timeout -k 10 300 python bench.py --mode train --gan --steps 10 --no-cpu-baseline > gpurun_out/r06/train_gan.json 2> gpurun_out/r06/train_gan.err; echo "gan rc=$?"
TGSR_GRAPH_G=0 timeout -k 10 300 python bench.py --mode train --steps 10 --no-cpu-baseline > gpurun_out/r06/train_g_eager.json 2> gpurun_out/r06/train_g_eager.err; echo "train eager rc=$?"
TGSR_GRAPH_G=0 timeout -k 10 300 python bench.py --mode train --gan --steps 10 --no-cpu-baseline > gpurun_out/r06/train_gan_eager.json 2> gpurun_out/r06/train_gan_eager.err; echo "gan eager rc=$?"
for f in train_g train_gan train_g_eager train_gan_eager; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/r06/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], "ms", d["value"], "img/s", "loss", d["final_loss"])
except Exception as e:
    print(f, "no line:", e); print(open('gpurun_out/r06/%s.err'%f).read()[-1500:])
PY
done
