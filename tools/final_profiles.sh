#!/bin/bash
# Final measurement set of a round (run on the GPU box through gpurun):  bash tools/final_profiles.sh <round tag>
# bench lines (default fp32, bf16 graph, f16 graph, bf16 eager lanes, bf16 B=8 graph, train) + rocprofv3 stats / PMC of
# the fp32 and bf16 paths.  Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -e
TAG=${1:-r05}
PART=${2:-all}      # all | bench | prof
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
if [ "$PART" != "prof" ]; then
python3 bench.py > $OUT/${TAG}_bench_fp32.json 2> $OUT/${TAG}_bench_fp32.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras none > $OUT/${TAG}_bench_fp32_driver_flags.json 2> $OUT/${TAG}_bench_fp32_driver_flags.err
python3 bench.py --eager --no-cpu-baseline --extras none > $OUT/${TAG}_bench_fp32_eager_lanes3.json 2> $OUT/${TAG}_bench_fp32_eager.err
echo "fp32 done"
python3 bench.py --dtype bf16 --graph > $OUT/${TAG}_bench_bf16_graph.json 2> $OUT/${TAG}_bench_bf16_graph.err
echo "bf16 graph done"
python3 bench.py --dtype f16 --graph --no-cpu-baseline > $OUT/${TAG}_bench_f16_graph.json 2> $OUT/${TAG}_bench_f16_graph.err
python3 bench.py --dtype bf16 --eager --no-cpu-baseline > $OUT/${TAG}_bench_bf16_eager_lanes3.json 2> $OUT/${TAG}_bench_bf16_eager.err
python3 bench.py --dtype bf16 --graph --batch 8 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_b8.json 2> $OUT/${TAG}_bench_bf16_graph_b8.err
python3 bench.py --dtype bf16 --graph --batch 64 --steps 10 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_b64.json 2> $OUT/${TAG}_bench_bf16_graph_b64.err
python3 bench.py --dtype bf16 --graph --graph-lanes 4 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_lanes4.json 2> $OUT/${TAG}_bench_bf16_graph_lanes4.err
python3 bench.py --dtype bf16 --graph --graph-lanes 8 --batch 8 --steps 48 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_b8_lanes8.json 2> $OUT/${TAG}_bench_bf16_graph_b8_lanes8.err
python3 bench.py --dtype bf16 --graph --batch 128 --steps 6 --no-cpu-baseline > $OUT/${TAG}_bench_bf16_graph_b128.json 2> $OUT/${TAG}_bench_bf16_graph_b128.err
echo "lp variants done"
python3 bench.py --mode train --steps 10 > $OUT/${TAG}_bench_train.json 2> $OUT/${TAG}_bench_train.err
python3 bench.py --mode train --gan --steps 10 > $OUT/${TAG}_bench_train_gan.json 2> $OUT/${TAG}_bench_train_gan.err
python3 bench.py --mode train --gan --damsm-encoder --steps 10 --no-cpu-baseline > $OUT/${TAG}_bench_train_gan_damsm.json 2> $OUT/${TAG}_bench_train_gan_damsm.err
echo "train done"
python3 bench.py --branch-num 5 --steps 12 --warmup 3 > $OUT/${TAG}_bench_x16_fp32.json 2> $OUT/${TAG}_bench_x16_fp32.err
python3 bench.py --branch-num 5 --dtype f16 --graph --steps 12 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_x16_f16_graph.json 2> $OUT/${TAG}_bench_x16_f16.err
python3 bench.py --branch-num 5 --dtype bf16 --graph --steps 12 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_x16_bf16_graph.json 2> $OUT/${TAG}_bench_x16_bf16.err
echo "x16 done"
TGSR_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 10 --no-cpu-baseline > $OUT/${TAG}_bench_rehearsal_2ranks.json 2> $OUT/${TAG}_bench_rehearsal.err || echo "rehearsal failed"
fi
if [ "$PART" != "bench" ]; then
# (counter passes incl. FETCH_SIZE / WRITE_SIZE for the train steps as well: `train.roofline.traffic`)
bash tools/profile_pmc.sh ${TAG}_train --mode train --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline
bash tools/profile_pmc.sh ${TAG}_train_gan --mode train --gan --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline
rm -f $OUT/${TAG}_train_*.log $OUT/${TAG}_train_gan_*.log
echo "train profiles done"
bash tools/profile_pmc.sh ${TAG}_fp32 --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --profile-every 0 --serial --extras none
bash tools/profile_pmc.sh ${TAG}_bf16 --dtype bf16 --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --profile-every 0 --serial --extras none
fi
for f in $OUT/${TAG}_bench_*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d.get('value_one_lane'), (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
" $f; done
