#!/usr/bin/env python3
"""In-kernel clock of lp_conv3x3_kernel's main loop on random data (MI355X_MICROARCH.md DVFS item 6): needs the
diagnostic build  VARIANTS=1024 bash tools/lp_conv_experiments.sh build ; then on the GPU box
  TGSR_LIB_PATH=$PWD/tgsr_amd/lib/dbg/libtgsr_dbg1024.so python tools/lp_conv_clock.py"""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import _lib, lp
L = ctypes.CDLL(_lib.LIB_PATH)
B, dev = 16, "cuda"
for cin, cout, H, glu in [(64, 128, 128, True), (64, 64, 128, False), (64, 128, 64, True)]:
    x = lp.from_nchw(torch.randn(B, cin, H, H, device=dev), "bf16")
    w = lp.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5), "bf16")
    co = cout // 2 if glu else cout
    out = lp.new_image(B, H, H, co, "bf16", dev)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    t0 = time.time()
    while time.time() - t0 < 2.0:                      # >= 2 s of back-to-back launches before the stamped one
        for _ in range(200):
            lp.conv3x3(x, w, cin, cout, sc, sh, glu=glu, out=out)
        torch.cuda.synchronize()
    n = min(4096, B * (H // 8) * (H // 32))
    buf = (ctypes.c_ulonglong * (10 * n))()
    assert L.tgsr_debug_read_lp_stamps(buf, 10 * n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 10).astype(np.int64)
    cyc, wall = a[:, :4], a[:, 4:8] * 10.0                           # shader cycles; ns (100 MHz counter)
    t0w = wall[:, 0].min()
    ghz = np.median((cyc[:, 2] - cyc[:, 1]) / np.maximum(wall[:, 2] - wall[:, 1], 1))
    med = lambda v: float(np.median(v)) / 1e3
    print("%d->%d @%d^2 (%d workgroups): in-kernel clock %.2f GHz; per workgroup, median us: entry->loop %.2f, main loop %.2f, "
          "epilogue %.2f; kernel span (first entry -> last exit) %.1f us" %
          (cin, cout, H, n, ghz, med(wall[:, 1] - wall[:, 0]), med(wall[:, 2] - wall[:, 1]), med(wall[:, 3] - wall[:, 2]),
           (wall[:, 3].max() - t0w) / 1e3))
    start = np.sort(wall[:, 0] - t0w) / 1e3
    end = np.sort(wall[:, 3] - t0w) / 1e3
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    print("    workgroup entry times (us after the first): 25%% %.1f  50%% %.1f  75%% %.1f  last %.1f;  exits: 25%% %.1f  50%% %.1f  "
          "75%% %.1f  last %.1f" % (q(start, .25), q(start, .5), q(start, .75), start[-1], q(end, .25), q(end, .5), q(end, .75), end[-1]))
    # where the workgroups ran: HW_ID (gfx9: cu_id bits 11:8, sh_id 12, se_id 15:13) and XCC_ID (bits 3:0)
    hw, xcc = a[:, 8], a[:, 9] & 0xF
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    first = wall[:, 0] - t0w < 2000.0                                # the first round
    by_cu = {}
    for i in np.nonzero(first)[0]:
        by_cu.setdefault(int(cu[i]), []).append(int(i))
    sizes = sorted(set(len(v) for v in by_cu.values()))
    print("    first round: %d workgroups on %d CUs (%s per CU); co-resident blockIdx pairs (first 12): %s; XCD of blockIdx 0..15: %s"
          % (int(first.sum()), len(by_cu), sizes, [tuple(v) for v in list(by_cu.values())[:12]], xcc[:16].tolist()))
    diffs = sorted(set(abs(v[1] - v[0]) for v in by_cu.values() if len(v) == 2))
    print("    |blockIdx difference| of co-residents: %s" % diffs[:20])
