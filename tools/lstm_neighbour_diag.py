#!/usr/bin/env python3
"""Diagnostic: WHAT changes inside the LSTM recurrence when lp_upconv_glu_kernel<.., 32, ..> runs beside it - its LDS-staged
pre-activations or its weight registers?  (LSTM built with -DTGSR_LSTM_CHECK into tgsr_amd/lib/diag/libtgsr_lcheck.so: the shipped kernel plus a check at its end; the
product library provides the neighbour)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tgsr_amd import lp
PRODUCT = os.environ.get("LIB") == "product"       # the shipped kernel itself (no integrity counts then)
L = ctypes.CDLL(os.path.join(ROOT, "tgsr_amd", "lib", "libtgsr_hip.so") if PRODUCT else os.path.join(ROOT, "tgsr_amd", "lib", "diag", "libtgsr_lcheck.so"))
vp, i32 = ctypes.c_void_p, ctypes.c_int
L.tgsr_bilstm_table_fwd.argtypes = [vp, i32, vp, i32, i32, vp, i32, vp, i32, vp, vp, vp]
B, T, H, ntok = 16, 18, 128, 41
dev = "cuda"
g = torch.Generator().manual_seed(0)
cap = torch.randint(1, ntok, (B, T), generator=g).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
table = (torch.randn(ntok, 2, 4 * H, generator=g) * 0.5).to(dev)
w_hh = (torch.randn(2, 4 * H, H, generator=g) * 0.08).to(dev)
words = torch.empty(B, 2 * H, T, device=dev); sent = torch.empty(B, 2 * H, device=dev)
def lstm():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.tgsr_bilstm_table_fwd(cap.data_ptr(), T, lens.data_ptr(), B, T, table.data_ptr(), ntok, w_hh.data_ptr(), H,
                                   words.data_ptr(), sent.data_ptr(), st) == 0
def counts():
    if PRODUCT:
        return 0, 0
    buf = (ctypes.c_uint * 2)()
    assert L.tgsr_debug_read_lcheck(buf) == 0
    return int(buf[0]), int(buf[1])
lstm(); torch.cuda.synchronize()
ref = words.clone()
c0 = counts()
R = lambda *sh: torch.randn(*sh, generator=g).to(dev)
x = lp.from_nchw(R(B, 32, 32, 32), "bf16")
wp = lp.pack_upconv_weight(R(64, 32, 3, 3) * 0.1, "bf16")
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
out = lp.new_image(B, 64, 64, 32, "bf16", dev)
side = torch.cuda.Stream()
bad = 0
for it in range(60):
    with torch.cuda.stream(side):
        for _ in range(3):
            lp.upconv_glu(x, wp, 32, 64, sc, sh, out=out)
    if os.environ.get("FRESH"):                      # fresh output tensors per launch, as ops.bilstm_table allocates them
        words = torch.empty(B, 2 * H, T, device=dev); sent = torch.empty(B, 2 * H, device=dev)
    lstm()
    with torch.cuda.stream(side):
        for _ in range(3):
            lp.upconv_glu(x, wp, 32, 64, sc, sh, out=out)
    torch.cuda.synchronize()
    bad += int(not torch.equal(words, ref))
c1 = counts()
print("LSTM launches that differ: %d of 60; wrong LDS pre-activation words %d -> %d, wrong weight registers %d -> %d"
      % (bad, c0[0], c1[0], c0[1], c1[1]))
