#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase cycles + in-kernel clock of the Winograd kernel (needs the -DTGSR_WINO_STAMPS build:
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTGSR_WINO_STAMPS -shared -o tgsr_amd/lib/diag/libtgsr_wstamps.so
      tgsr_amd/csrc/tgsr_winograd.hip tgsr_amd/csrc/tgsr_misc.hip)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "tgsr_amd", "lib", "diag", "libtgsr_wstamps%s.so" % os.environ.get("WEXP", "")))
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
L.tgsr_wino_conv3x3_fwd.argtypes = [vp, i64, i32, i32, i32, i32, vp, i32, vp, vp, vp, i64, vp, i64, i32, vp]
L.tgsr_pack_wino_weight.argtypes = [vp, vp, i32, i32, i32, vp]
L.tgsr_packed_wino_weight_elems.restype = i64
def run(B, cin, cout, h, glu, res):
    dev = "cuda"
    x = torch.randn(B, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    wp = torch.empty(L.tgsr_packed_wino_weight_elems(cout, cin), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.tgsr_pack_wino_weight(w.data_ptr(), wp.data_ptr(), cout, cin, glu, st) == 0
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev) * 0.1
    co = cout // 2 if glu else cout
    r = torch.randn(B, co, h, h, device=dev) if res else None
    out = torch.empty(B, co, h, h, device=dev)
    def go():
        rc = L.tgsr_wino_conv3x3_fwd(x.data_ptr(), cin * h * h, B, cin, h, h, wp.data_ptr(), cout, sc.data_ptr(), sh.data_ptr(),
                                     r.data_ptr() if res else None, co * h * h, out.data_ptr(), co * h * h, 1 if glu else 0, st)
        assert rc == 0
    for _ in range(20): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    n = 8 * 8192
    buf = (ctypes.c_ulonglong * n)()
    assert L.tgsr_debug_read_wstamps(buf, n) == 0
    nwg = B * ((h + 31) // 32) * ((h + 3) // 4) * (cout // 64)
    s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)[:min(nwg, 8192)]
    s = s[(s[:, 3] > s[:, 0]) & (s[:, 7] > s[:, 6])]
    clk = (s[:, 3] - s[:, 0]) / np.maximum(1, (s[:, 7] - s[:, 6])) * 100e6
    pro, main, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    rt0 = s[:, 6].min(); span = (s[:, 7].max() - rt0) * 10
    st_ns = (s[:, 6] - rt0) * 10; en_ns = (s[:, 7] - rt0) * 10
    print("WINO B%d %d->%d @%d glu%d res%d: %.1f us, %d WGs; clock %.3f GHz; cycles/WG: prologue %d  main %d (%d/stage)  epilogue %d ; span %d ns"
          % (B, cin, cout, h, glu, res, e0.elapsed_time(e1) * 1e3, len(s), np.median(clk) / 1e9, np.median(pro), np.median(main),
             np.median(main) / ((cin + 3) // 4), np.median(epi), span))
    print("   WG start ns pct 10/50/90/100: %s   end ns pct 10/50/90/100: %s" % (
        np.percentile(st_ns, [10, 50, 90, 100]).astype(int), np.percentile(en_ns, [10, 50, 90, 100]).astype(int)))
if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    if a: run(*a)
    else:
        run(16, 64, 128, 128, 1, 0); run(16, 64, 64, 128, 0, 1); run(16, 64, 128, 64, 1, 0)
