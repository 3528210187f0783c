#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase cycles + in-kernel clock of the F(4x4,3x3) kernel (builds its own -DTGSR_WINO4_STAMPS library
under tgsr_amd/lib/diag/ when missing).   python tools/wino4_stamps.py [B cin cout h glu res]
W4EXP=<bits> builds a timing-experiment variant with parts of the stage compiled out (1: no U copies, 2: no transform, 4: no raw
copies; wrong results by construction, only the times mean anything)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tgsr_amd", "lib", "diag", "libtgsr_w4stamps%s.so" % os.environ.get("W4EXP", ""))
if not os.path.exists(SO):
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize",
                           "-DTGSR_WINO4_STAMPS", "-shared", "-o", SO, os.path.join(ROOT, "tgsr_amd/csrc/tgsr_winograd4.hip"),
                           os.path.join(ROOT, "tgsr_amd/csrc/tgsr_misc.hip")] + (["-DTGSR_W4_EXP=" + os.environ["W4EXP"]] if os.environ.get("W4EXP") else []))
L = ctypes.CDLL(SO)
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
L.tgsr_wino4_conv3x3_fwd.argtypes = [vp, i64, i32, i32, i32, i32, vp, i32, vp, vp, vp, i64, vp, i64, i32, vp]
L.tgsr_pack_wino4_weight.argtypes = [vp, vp, i32, i32, i32, vp]
L.tgsr_packed_wino4_weight_elems.restype = i64
def run(B, cin, cout, h, glu, res):
    dev = "cuda"
    x = torch.randn(B, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    wp = torch.empty(L.tgsr_packed_wino4_weight_elems(cout, cin), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.tgsr_pack_wino4_weight(w.data_ptr(), wp.data_ptr(), cout, cin, glu, st) == 0
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev) * 0.1
    co = cout // 2 if glu else cout
    r = torch.randn(B, co, h, h, device=dev) if res else None
    out = torch.empty(B, co, h, h, device=dev)
    def go():
        rc = L.tgsr_wino4_conv3x3_fwd(x.data_ptr(), cin * h * h, B, cin, h, h, wp.data_ptr(), cout, sc.data_ptr(), sh.data_ptr(),
                                      r.data_ptr() if res else None, co * h * h, out.data_ptr(), co * h * h, 1 if glu else 0, st)
        assert rc == 0
    for _ in range(20): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    n = 8 * 8192
    buf = (ctypes.c_ulonglong * n)()
    assert L.tgsr_debug_read_w4stamps(buf, n) == 0
    nwg = B * ((h + 63) // 64) * ((h + 7) // 8) * (cout // 64)
    s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)[:min(nwg, 8192)]
    s = s[(s[:, 3] > s[:, 0]) & (s[:, 7] > s[:, 6])]
    clk = (s[:, 3] - s[:, 0]) / np.maximum(1, (s[:, 7] - s[:, 6])) * 100e6
    pro, main, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    rt0 = s[:, 6].min(); span = (s[:, 7].max() - rt0) * 10
    st_ns = (s[:, 6] - rt0) * 10; en_ns = (s[:, 7] - rt0) * 10
    print("WINO4 B%d %d->%d @%d glu%d res%d: %.1f us, %d WGs; clock %.3f GHz; cycles/WG: prologue %d  main %d (%d/stage)  epilogue %d ; span %d ns"
          % (B, cin, cout, h, glu, res, e0.elapsed_time(e1) * 1e3, len(s), np.median(clk) / 1e9, np.median(pro), np.median(main),
             np.median(main) / ((cin + 3) // 4), np.median(epi), span))
    print("   WG start ns pct 10/50/90/100: %s   end ns pct 10/50/90/100: %s" % (
        np.percentile(st_ns, [10, 50, 90, 100]).astype(int), np.percentile(en_ns, [10, 50, 90, 100]).astype(int)))
if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:7]]
    if a: run(*a)
    else:
        run(16, 64, 128, 128, 1, 0); run(16, 64, 64, 128, 0, 1); run(16, 128, 128, 128, 1, 0)
