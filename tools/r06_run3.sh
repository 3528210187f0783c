#!/bin/bash
# round 6: the closed boundary stubs + graph tests + the default bench line with the DAMSM-encoder run in its train object
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_variants.py tests/test_hip_gan.py tests/test_hip_dp.py tests/test_hip_train.py -x -q -k "variants or graph_replayed or gloo or conv_bn_leaky_block or single_rank or pack_cache" > $OUT/t3.log 2>&1
echo "pytest rc=$?"; tail -n 25 $OUT/t3.log
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r06/bench_default.json').read().strip().splitlines()[-1])
    print("headline", d["value"], d["ms_per_step"])
    for r in d["train"]["runs"]:
        print(r.get("ms_per_step"), r.get("value"), r.get("error"), json.dumps(r.get("device_time"))[:600])
    print(json.dumps(d.get("ranks")))
except Exception as e:
    print("no line", e); print(open('gpurun_out/r06/bench_default.err').read()[-3000:])
PY
