#!/bin/bash
# The full GPU suite and smoke() on the tree as it stands.
set -o pipefail
OUT=gpurun_out/r06
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/gputest_full2.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -n 6 $OUT/gputest_full2.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke2.log 2>&1; echo "smoke rc=$?"; tail -n 3 $OUT/smoke2.log
exit $rc
