#!/bin/bash
# Flat Adam: parity with torch.optim.Adam, the train / gan / dp suites on it, the three train lines.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests/test_hip_optim.py tests/test_hip_custom_ops.py tests/test_hip_parity.py tests/test_hip_train.py tests/test_hip_gan.py tests/test_hip_dp.py tests/test_hip_variants.py -x -q > $OUT/t12.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 12 $OUT/t12.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for cfgname in "g:" "gan:--gan" "enc:--gan --damsm-encoder"; do
  name=${cfgname%%:*}; flags=${cfgname#*:}
  timeout -k 10 400 python bench.py --mode train $flags --steps 10 --no-cpu-baseline > $OUT/train_${name}_flat.json 2> $OUT/train_${name}_flat.err; echo "$name rc=$?"
done
TGSR_FLAT_ADAM=0 TGSR_BN_FUSE_SMALL=0 timeout -k 10 400 python bench.py --mode train --gan --steps 10 --no-cpu-baseline > $OUT/train_gan_torchadam.json 2> $OUT/train_gan_torchadam.err; echo "gan torch adam, two-pass small BN rc=$?"
python - $OUT/train_g_flat.json $OUT/train_gan_flat.json $OUT/train_enc_flat.json $OUT/train_gan_torchadam.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); dt=d.get("device_time") or {}
        print(f, d["ms_per_step"], d["value"], d.get("graph_policy"), "non_tgsr", dt.get("non_tgsr_share"), dt.get("non_tgsr_ms"), dt.get("kernel_ms"), dt.get("largest_non_tgsr"))
    except Exception as e: print("no line", f, e)
PY
