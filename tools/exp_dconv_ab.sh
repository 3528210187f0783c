#!/bin/bash
# A/B of the discriminators' GEMMs: fp32 MFMA (0) vs three-piece bf16 operands (1; 3 = with pre-split weights).  profiles/HISTORY.md 3.18.
#   tools/exp_dconv_ab.sh [B] [modes...]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B=${1:-32}
shift || true
for m in ${@:-0 1 3}; do
  TGSR_DCONV_SPLIT=$m python3 tools/exp_dconv.py $B > gpurun_out/dconv_layers_mode$m.txt
  echo "== TGSR_DCONV_SPLIT=$m"; cat gpurun_out/dconv_layers_mode$m.txt
done
