#!/bin/bash
# Round-6 re-entry: the full GPU suite on the restored tree, then the default bench line.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputest_full.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -n 8 $OUT/gputest_full.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench_default.json
