#!/usr/bin/env python3
"""What would fusing the fp32 image heads into their upBlocks (profiles/HISTORY.md 3.11) remove from the PRODUCER?  Times the two 256^2
upBlocks of a forward (upwino_kernel: G_SR_NET_low 64 -> 64 GLU, NetG_highweight 32 -> 64 GLU, batch 16, 128^2 -> 256^2) with the
shipped library and with a diagnostic build whose epilogue computes everything but does not store
(-DTGSR_UPW_NOSTORE: results wrong by construction, only the times mean anything), and the two stand-alone heads that read them.
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DTGSR_UPW_NOSTORE -shared \
        -o tgsr_amd/lib/diag/libtgsr_upw_nostore.so tgsr_amd/csrc/tgsr_upwino.hip tgsr_amd/csrc/tgsr_misc.hip"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgsr_amd import ops, _lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = ctypes.CDLL(os.path.join(ROOT, "tgsr_amd", "lib", "diag", "libtgsr_upw_nostore.so"))
sig = _lib.SIGNATURES["tgsr_upwino_glu_fwd"]
D.tgsr_upwino_glu_fwd.restype, D.tgsr_upwino_glu_fwd.argtypes = sig
S = _lib.lib().tgsr_upwino_glu_fwd
dev, B = "cuda", 16


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, cin in (("G_SR_NET_low h_net3.upsample 64->64(GLU 32)", 64), ("NetG_highweight upscale8x 32->64(GLU 32)", 32)):
    x = torch.randn(B, cin, 128, 128, device=dev)
    w = torch.randn(64, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    up = ops.pack_upwino_weight(w, True)
    sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    out = torch.empty(B, 32, 256, 256, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    def call(fn):
        rc = fn(p(x), cin * 128 * 128, B, cin, 128, 128, p(up), 64, p(sc), p(sh), p(out), 32 * 256 * 256, st)
        assert rc == 0, rc
    t_full, t_nost = timeit(lambda: call(S)), timeit(lambda: call(D.tgsr_upwino_glu_fwd))
    print("%s: %.1f us with its 134 MB of stores, %.1f us without them" % (name, t_full, t_nost))
h = torch.randn(B, 32, 256, 256, device=dev)
w3, w5, add = torch.randn(3, 32, 3, 3, device=dev) * 0.1, torch.randn(3, 32, 5, 5, device=dev) * 0.1, torch.randn(B, 3, 256, 256, device=dev)
print("stand-alone 3x3 head @256^2: %.1f us; 5x5 + tanh + a*SRb head: %.1f us" % (
    timeit(lambda: ops.conv_to3(h, w3)), timeit(lambda: ops.conv_to3(h, w5, tanh_axpy=True, addend=add, alpha=0.5))))
