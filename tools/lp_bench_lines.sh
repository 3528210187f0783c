set -e
TAG=r02i
OUT=gpurun_out
python3 bench.py --dtype bf16 --graph > $OUT/${TAG}_bench_bf16_graph.json 2>/dev/null
python3 bench.py --dtype f16 --graph --no-cpu-baseline > $OUT/${TAG}_bench_f16_graph.json 2>/dev/null
python3 bench.py --dtype bf16 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_lanes.json 2>/dev/null
python3 bench.py --dtype bf16 --graph --batch 8 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_b8.json 2>/dev/null
python3 bench.py --dtype bf16 --graph --batch 64 --steps 10 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_b64.json 2>/dev/null
python3 bench.py --dtype bf16 --graph --graph-lanes 4 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_lanes4.json 2>/dev/null
python3 bench.py --dtype bf16 --graph --graph-lanes 8 --batch 8 --steps 48 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_b8_lanes8.json 2>/dev/null
python3 bench.py --dtype bf16 --graph --batch 128 --steps 6 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_b128.json 2>/dev/null
for f in $OUT/${TAG}_bench_bf16*.json $OUT/${TAG}_bench_f16*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['bound'], r['frac'], r.get('bytes_from'), r.get('mfma_frac'), d['step_roofline']['hbm_frac'])
" $f; done
