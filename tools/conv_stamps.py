#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase cycles + in-kernel clock of the conv kernel (needs the -DTGSR_CONV_STAMPS build:
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTGSR_CONV_STAMPS -shared -o tgsr_amd/lib/diag/libtgsr_stamps.so
      tgsr_amd/csrc/tgsr_conv3x3.hip tgsr_amd/csrc/tgsr_misc.hip)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "tgsr_amd", "lib", "diag", "libtgsr_stamps.so"))
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
L.tgsr_conv3x3_fwd.argtypes = [vp, i64, i32, i32, i32, i32, vp, i32, vp, vp, vp, i64, vp, i64, i32, i32, vp]
L.tgsr_pack_conv_weight.argtypes = [vp, vp, i32, i32, i32, vp]
L.tgsr_packed_weight_elems.restype = i64
def run(B, cin, cout, h, glu, up, res):
    dev = "cuda"
    x = torch.randn(B, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    wp = torch.empty(L.tgsr_packed_weight_elems(cout, cin, 3), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.tgsr_pack_conv_weight(x.new_tensor([]).data_ptr() or w.data_ptr(), wp.data_ptr(), cout, cin, 3, st) if False else L.tgsr_pack_conv_weight(w.data_ptr(), wp.data_ptr(), cout, cin, 3, st)
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev) * 0.1
    ho = 2 * h if up else h; co = cout // 2 if glu else cout
    r = torch.randn(B, co, ho, ho, device=dev) if res else None
    out = torch.empty(B, co, ho, ho, device=dev)
    def go():
        rc = L.tgsr_conv3x3_fwd(x.data_ptr(), cin * h * h, B, cin, h, h, wp.data_ptr(), cout, sc.data_ptr(), sh.data_ptr(),
                                r.data_ptr() if res else None, co * ho * ho, out.data_ptr(), co * ho * ho, 1 if glu else 0, up, st)
        assert rc == 0
    for _ in range(20): go()           # settle the clock
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    n = 8 * 8192
    buf = (ctypes.c_ulonglong * n)()
    assert L.tgsr_debug_read_stamps(buf, n) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
    s = s[s[:, 3] > 0]
    s = s[: B * ((2 * h if up else h) // 4 + 1) * ((2 * h if up else h) // 32 + 1)]   # drop stale rows of earlier launches
    s = s[(s[:, 3] > s[:, 0]) & (s[:, 7] > s[:, 6])]
    clk = (s[:, 3] - s[:, 0]) / np.maximum(1, (s[:, 7] - s[:, 6])) * 100e6
    pro, main, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    rt0 = s[:, 6].min(); span = (s[:, 7].max() - rt0) * 10   # ns (100 MHz constant clock)
    st_ns = (s[:, 6] - rt0) * 10; en_ns = (s[:, 7] - rt0) * 10
    t0 = 0
    print("B%d %d->%d @%d glu%d up%d res%d: %.1f us, %d WGs; clock %.3f GHz (median); cycles/WG: prologue %d  main %d  epilogue %d ;"
          " kernel span %d cycles; WG start spread: %d..%d" % (B, cin, cout, h, glu, up, res, e0.elapsed_time(e1) * 1e3, len(s),
          np.median(clk) / 1e9, np.median(pro), np.median(main), np.median(epi), span, st_ns.min(), np.percentile(st_ns, 50)))
    print("   WG start ns pct 10/50/90/100: %s   end ns pct 10/50/90/100: %s  (kernel span %d ns)" % (
        np.percentile(st_ns, [10, 50, 90, 100]).astype(int), np.percentile(en_ns, [10, 50, 90, 100]).astype(int), span))
if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    if a: run(*a)
    else:
        run(16, 64, 128, 128, 1, 0, 0); run(16, 64, 64, 128, 0, 0, 1); run(16, 64, 128, 64, 1, 0, 0)
