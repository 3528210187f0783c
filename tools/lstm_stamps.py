#!/usr/bin/env python3
"""Diagnostic: where a step of the LSTM recurrence spends its time (needs the stamped build:
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DTGSR_LSTM_STAMPS -shared -o tgsr_amd/lib/diag/libtgsr_lstamps.so
      tgsr_amd/csrc/tgsr_lstm.hip tgsr_amd/csrc/tgsr_misc.hip).  Prints per-phase cycles of thread 0 (median over workgroups and steps) and the shader clock."""
import ctypes, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "tgsr_amd", "lib", "diag", "libtgsr_lstamps.so"))
vp, i32 = ctypes.c_void_p, ctypes.c_int
L.tgsr_bilstm_table_fwd.argtypes = [vp, i32, vp, i32, i32, vp, i32, vp, i32, vp, vp, vp]
B, T, H, ntok = 16, 18, 128, 41
dev = "cuda"
cap = torch.randint(1, ntok, (B, T), device=dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
table = torch.randn(ntok, 2, 4 * H, device=dev) * 0.1
w_hh = torch.randn(2, 4 * H, H, device=dev) * 0.05
words = torch.empty(B, 2 * H, T, device=dev); sent = torch.empty(B, 2 * H, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def go():
    rc = L.tgsr_bilstm_table_fwd(cap.data_ptr(), T, lens.data_ptr(), B, T, table.data_ptr(), ntok, w_hh.data_ptr(), H,
                                 words.data_ptr(), sent.data_ptr(), st)
    assert rc == 0, rc
for _ in range(10): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); go(); e1.record(); torch.cuda.synchronize()
n = 64 * 128
buf = (ctypes.c_ulonglong * n)()
assert L.tgsr_debug_read_lstamps(buf, n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(64, 128).astype(np.int64)[:2 * B]
tot = s[:, 4 + 3 * (T - 1)] - s[:, 0]
clk = tot / np.maximum(1, s[:, 127] - s[:, 126]) * 100e6
ph = np.stack([s[:, 2 + 3 * k: 5 + 3 * k] - s[:, 1 + 3 * k: 4 + 3 * k] for k in range(1, T)], 0)   # [T-1][wg][3]
print("LSTM B%d T%d H%d: %.1f us; shader clock %.3f GHz; prologue %d cycles; per step (median, cycles): mat-vec %d  exchange + gates %d  barrier %d  = %d"
      % (B, T, H, e0.elapsed_time(e1) * 1e3, np.median(clk) / 1e9, np.median(s[:, 1] - s[:, 0]), *np.median(ph, (0, 1)).astype(int),
         int(np.median(ph.sum(2)))))
