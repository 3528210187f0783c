#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
TGSR_LIB_PATH=$PWD/tgsr_amd/lib/libtgsr_hip_oldbn.so python tools/bench_bn.py > $OUT/bench_bn_old.txt 2>&1; cat $OUT/bench_bn_old.txt
python tools/bench_bn.py > $OUT/bench_bn_new.txt 2>&1; cat $OUT/bench_bn_new.txt
timeout -k 10 900 python -m pytest tests/test_hip_variants.py tests/test_hip_gan.py tests/test_hip_dp.py tests/test_hip_train.py -x -q > $OUT/t4.log 2>&1
echo "pytest rc=$?"; tail -n 25 $OUT/t4.log
for m in "" "--gan"; do timeout -k 10 300 python bench.py --mode train $m --steps 10 --no-cpu-baseline > $OUT/train_bn$m.json 2> $OUT/train_bn$m.err; python - "$OUT/train_bn$m.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])
except Exception as e: print("no line", e)
PY
done
