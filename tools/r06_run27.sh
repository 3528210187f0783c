#!/bin/bash
# Hardware queues below ROCclr's default of 4 (8 was twice as slow on the train steps: cross-queue dependencies).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
for q in 4 2 1; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 400 python bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/hq${q}_enc.json 2> $OUT/hq${q}_enc.err; echo "q=$q enc rc=$?"
  timeout -k 10 400 python bench.py --mode train --gan --steps 10 --no-cpu-baseline > $OUT/hq${q}_gan.json 2> $OUT/hq${q}_gan.err; echo "q=$q gan rc=$?"
  timeout -k 10 400 python bench.py --mode train --steps 10 --no-cpu-baseline > $OUT/hq${q}_g.json 2> $OUT/hq${q}_g.err; echo "q=$q g rc=$?"
  timeout -k 10 400 python bench.py --no-cpu-baseline --extras none > $OUT/hq${q}_infer.json 2> $OUT/hq${q}_infer.err; echo "q=$q infer rc=$?"
  timeout -k 10 400 python bench.py --no-cpu-baseline --extras none --dtype bf16 --graph > $OUT/hq${q}_bf16.json 2> $OUT/hq${q}_bf16.err; echo "q=$q bf16 rc=$?"
done
python - $OUT <<'PY'
import json,sys,os
for q in (4,2,1):
    for n in ("enc","gan","g","infer","bf16"):
        f=os.path.join(sys.argv[1],"hq%d_%s.json"%(q,n))
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); print(q, n, d["ms_per_step"], d["value"], d.get("value_throughput_form"), (d.get("graph_policy") or {}).get("chosen"))
        except Exception as e: print("no line", f, e)
PY
