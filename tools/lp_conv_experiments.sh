#!/bin/bash
# Timing decomposition of lp_conv3x3_kernel: builds variants of the library from the DIAGNOSTIC copy of the kernel source
# (tools/diag/tgsr_lp_conv_dbg.hip; the shipped tgsr_amd/csrc/tgsr_lp_conv.hip carries none of this) with parts switched off
# (-DLP_DBG bits: 1 no weight LDS-DMA after the prologue, 2 no output stores, 4 no B-fragment ds_reads, 8 no A-fragment
# ds_reads, 16 no tile LDS-DMA, 64 the same FLOPs as 16x16x32 MFMAs, 128 | 256 | 512 a 3 us head start for half of
# the workgroups) and, on a GPU box, times the generator's layer shapes with each.  Results are WRONG by
# construction; only the times mean anything.   build here:  bash tools/lp_conv_experiments.sh build
#                                               on the box:  bash tools/lp_conv_experiments.sh run > gpurun_out/lp_dbg.txt
set -e
cd "$(dirname "$0")/.."
VARIANTS="${VARIANTS:-0 1 2 4 8 12 13 15 31}"
if [ "$1" = build ]; then
  make -C tgsr_amd/csrc -j8 >/dev/null
  mkdir -p tgsr_amd/lib/dbg
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLP_DBG=$v \
      -Itgsr_amd/csrc -c tools/diag/tgsr_lp_conv_dbg.hip -o tgsr_amd/lib/dbg/lp_conv_$v.o &
  done
  wait
  for v in $VARIANTS; do
    objs=$(ls tgsr_amd/lib/obj/*.o | grep -v tgsr_lp_conv.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tgsr_amd/lib/dbg/libtgsr_dbg$v.so $objs tgsr_amd/lib/dbg/lp_conv_$v.o
  done
  rm -f tgsr_amd/lib/dbg/*.o
  ls -la tgsr_amd/lib/dbg
else
  for v in $VARIANTS; do
    echo "=== LP_DBG=$v"
    TGSR_LIB_PATH=$PWD/tgsr_amd/lib/dbg/libtgsr_dbg$v.so python tools/bench_lp_conv.py --layers "${LAYERS:-6,7,3,9,10,15}" --reps 50
  done
fi
