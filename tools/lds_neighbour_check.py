#!/usr/bin/env python3
"""Diagnostic: does a kernel's result change when ANOTHER kernel runs beside it on the same CUs?
The LSTM recurrence (its per-step inputs live in LDS for the whole launch) is replayed on one stream while a candidate
neighbour kernel loops on a second stream; every output is compared with the one the recurrence gives alone.
    python3 tools/lds_neighbour_check.py [iterations]          (ONLY=upconv,glu selects neighbours, NO_CANARY=1 skips the canaries)
The canaries (LDS contents, registers, packed FMAs on registers, packed FMAs fed from LDS, plain LDS write/barrier/read) live in
tools/diag/lds_canary.hip:
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -shared -o tgsr_amd/lib/diag/liblds_canary.so tools/diag/lds_canary.hip
What it found is profiles/HISTORY.md section 3.13: v_pk_fma_f32 fed from ds_read_b128 gives other results beside MFMA-bound kernels."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tgsr_amd import ops, lp  # noqa: E402

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(0)
B, T, H, ntok = 16, 18, 128, 41
cap = torch.randint(1, ntok, (B, T), generator=g).to(dev)
lens = [T] * B
table = (torch.randn(ntok, 2, 4 * H, generator=g) * 0.5).to(dev)
w_hh = (torch.randn(2, 4 * H, H, generator=g) * 0.08).to(dev)


def lstm():
    return ops.bilstm_table(cap, lens, table, w_hh)


ref_w, ref_s = lstm()
torch.cuda.synchronize()
for _ in range(20):
    w, s_ = lstm()
    assert torch.equal(w, ref_w) and torch.equal(s_, ref_s), "the recurrence is not reproducible on its own"

dt = "bf16"
R = lambda *sh: torch.randn(*sh, generator=g).to(dev)  # noqa: E731


def mk_conv(cin, cout, hw, glu, res):
    x = lp.from_nchw(R(B, cin, hw, hw), dt)
    wp = lp.pack_conv3x3_weight(R(cout, cin, 3, 3) * 0.1, dt)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    co = cout // 2 if glu else cout
    out = lp.new_image(B, hw, hw, co, dt, dev)
    rs = lp.from_nchw(R(B, co, hw, hw), dt) if res else None
    return lambda: lp.conv3x3(x, wp, cin, cout, sc, sh, glu=glu, residual=rs, out=out)


def mk_upconv(cin, hw, K):
    x = lp.from_nchw(R(B, cin, hw, hw), dt)
    wp = lp.pack_upconv_weight(R(64, cin, 3, 3) * 0.1, dt)
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    out = lp.new_image(B, 2 * hw, 2 * hw, 32, dt, dev)
    if K == 0:
        return lambda: lp.upconv_glu(x, wp, cin, 64, sc, sh, out=out)
    hwp = lp.pack_to3_weight(R(3, 32, K, K) * 0.1, dt)
    part = torch.empty(lp.head_partial_elems(B, 2 * hw, 2 * hw, K), device=dev)
    return lambda: lp.upconv_glu_head(x, wp, cin, 64, sc, sh, hwp, K, partial=part, out=out)


def mk_stem(hw):
    x = R(B, 3, hw, hw)
    w = R(64, 3, 3, 3) * 0.2
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    out = lp.new_image(B, hw, hw, 32, dt, dev)
    return lambda: lp.stem(x, w, sc, sh, out=out)


def mk_fp32_wino(cin, cout, hw, glu):
    x = R(B, cin, hw, hw)
    up = ops.pack_wino_weight(R(cout, cin, 3, 3) * 0.1, glu, False)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    return lambda: ops.conv3x3_wino(x, up, cout, sc, sh, glu, None)


cands = {
    "nothing": None,
    "lp_stem 32^2": mk_stem(32),
    "lp_conv 32->64 glu @32^2": mk_conv(32, 64, 32, True, False),
    "lp_conv 32->32 +res @32^2": mk_conv(32, 32, 32, False, True),
    "lp_conv 64->128 glu @32^2": mk_conv(64, 128, 32, True, False),
    "lp_conv 64->128 glu @128^2": mk_conv(64, 128, 128, True, False),
    "lp_upconv 32 @32->64": mk_upconv(32, 32, 0),
    "lp_upconv+head5 32 @32->64": mk_upconv(32, 32, 5),
    "lp_upconv+head3 64 @64->128": mk_upconv(64, 64, 3),
    "fp32 wino 64->128 glu @64^2": mk_fp32_wino(64, 128, 64, True),
}
import ctypes
L = ctypes.CDLL(os.path.join(ROOT, "tgsr_amd", "lib", "diag", "liblds_canary.so"))
L.lds_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
_rep = torch.zeros(4 + 4 * 64, dtype=torch.int32, device=dev)
cands["idle canary 16 KB (timing only)"] = lambda: L.lds_canary(1024, 16, 1500, _rep.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
cands["idle canary 60 KB (timing only)"] = lambda: L.lds_canary(512, 60, 1500, _rep.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
only = os.environ.get("ONLY")
if only:
    cands = {k: v for k, v in cands.items() if v is None or any(o in k for o in only.split(","))}
side = torch.cuda.Stream()
for name, fn in cands.items():
    bad = 0
    torch.cuda.synchronize()
    for it in range(N):
        if fn is not None:
            with torch.cuda.stream(side):
                for _ in range(3):
                    fn()
        w, s_ = lstm()
        if fn is not None:
            with torch.cuda.stream(side):
                for _ in range(3):
                    fn()
        torch.cuda.synchronize()
        if not (torch.equal(w, ref_w) and torch.equal(s_, ref_s)):
            bad += 1
    print("%-32s: %d of %d LSTM launches differ" % (name, bad, N), flush=True)


# ---- the same neighbours against an LDS canary (tools/diag/lds_canary.hip): which words of a foreign allocation change?
L.lds_canary_words.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for name, fn in cands.items():
    if fn is None or "canary" in name or os.environ.get("NO_CANARY"):
        continue
    for words in (4096, 4146, 4224, 12288, 12568, 640):
        rep = torch.zeros(4 + 4 * 64, dtype=torch.int32, device=dev)
        rep[1] = 0x7fffffff
        torch.cuda.synchronize()
        for it in range(20):
            assert L.lds_canary_words(512, words, 3000, 0, rep.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            with torch.cuda.stream(side):
                for _ in range(6):
                    fn()
        torch.cuda.synchronize()
        r = rep.cpu().numpy().astype("uint32")
        msg = "%-32s canary %6d B: %d corrupted words" % (name, words * 4, r[0])
        if r[0]:
            ent = [(int(r[4 + 4 * k]), int(r[5 + 4 * k]), hex(int(r[6 + 4 * k])), hex(int(r[7 + 4 * k]))) for k in range(min(int(r[0]), 10))]
            msg += "; byte range [%d, %d]; (block, word, got, want) %s" % (r[1] * 4, r[2] * 4 + 3, ent[:3])
        print(msg, flush=True)

# ---- the same against a register canary (150 VGPRs x 512 threads / 200 x 256 hold a pattern while the neighbour runs)
L.vgpr_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
for name, fn in cands.items():
    if fn is None or "canary" in name or os.environ.get("NO_CANARY"):
        continue
    for thr in (512, 256):
        rep = torch.zeros(4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for it in range(20):
            assert L.vgpr_canary(512, thr, 3000, rep.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            with torch.cuda.stream(side):
                for _ in range(6):
                    fn()
        torch.cuda.synchronize()
        print("%-32s register canary %d threads: %d corrupted registers" % (name, thr, int(rep[0])), flush=True)

# ---- and a packed-FMA canary: v_pk_fma_f32 against v_fma_f32 on the same data, in registers only
L.pkfma_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for name, fn in cands.items():
    if os.environ.get("NO_CANARY") or (fn is not None and "canary" in name):
        continue
    for thr in (512, 256):
        rep = torch.zeros(4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for it in range(20):
            assert L.pkfma_canary(512, thr, 400, rep.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            if fn is not None:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        fn()
        torch.cuda.synchronize()
        print("%-32s packed-FMA canary %d threads: %d results differ from v_fma_f32" % (name, thr, int(rep[0])), flush=True)

# ---- packed FMAs fed from LDS (the LSTM recurrence's inner loop in isolation)
L.pkfma_lds_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for name, fn in cands.items():
    if os.environ.get("NO_CANARY") or (fn is not None and "canary" in name):
        continue
    for big in (1, 0, 2, 3):
        rep = torch.zeros(4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for it in range(20):
            assert L.pkfma_lds_canary(64, big, 60, rep.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            if fn is not None:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        fn()
        torch.cuda.synchronize()
        print("%-32s packed-FMA-from-LDS canary (%s LDS): %d results differ" % (name, {1: "49 KB", 0: "3 KB", 2: "3 KB, loads in registers of their own", 3: "3 KB, s_nop x32 before the FMAs"}[big], int(rep[0])), flush=True)

# ---- plain LDS write -> barrier -> read back (no arithmetic)
L.lds_read_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for name, fn in cands.items():
    if os.environ.get("NO_CANARY") or (fn is not None and "canary" in name):
        continue
    for mode in (0, 1):
        rep = torch.zeros(4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for it in range(20):
            assert L.lds_read_canary(64, mode, 60, rep.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            if fn is not None:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        fn()
        torch.cuda.synchronize()
        print("%-32s LDS write/barrier/read canary (%s): %d values differ" % (name, "ds_read_b128" if mode == 0 else "ds_read_b32", int(rep[0])), flush=True)
