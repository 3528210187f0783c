#!/bin/bash
# The committed lines once more with the round's own counter tables in profiles/ (counters_from / traffic_from: r06_*).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
python3 bench.py > $OUT/r06_bench_fp32.json 2> $OUT/r06_bench_fp32.err; echo "fp32 rc=$?"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras none > $OUT/r06_bench_fp32_driver_flags.json 2> /dev/null; echo "driver flags rc=$?"
python3 bench.py --dtype bf16 --graph > $OUT/r06_bench_bf16_graph.json 2> /dev/null; echo "bf16 rc=$?"
python3 bench.py --mode train --steps 10 > $OUT/r06_bench_train.json 2> /dev/null; echo "train rc=$?"
python3 bench.py --mode train --gan --steps 10 > $OUT/r06_bench_train_gan.json 2> /dev/null; echo "gan rc=$?"
python3 bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/r06_bench_train_gan_damsm.json 2> /dev/null; echo "enc rc=$?"
TGSR_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 10 --no-cpu-baseline > $OUT/r06_bench_rehearsal_2ranks.json 2> $OUT/r06_bench_rehearsal.err || echo "rehearsal failed"
python3 - $OUT <<'PY'
import json,sys,os
for n in ("fp32","fp32_driver_flags","bf16_graph","train","train_gan","train_gan_damsm","rehearsal_2ranks"):
    f=os.path.join(sys.argv[1],"r06_bench_%s.json"%n)
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
        print(n, d["value"], d["ms_per_step"], r.get("frac"), r.get("counters_from") or r.get("traffic_from"))
    except Exception as e: print("unreadable", f, e)
PY
