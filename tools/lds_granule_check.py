#!/usr/bin/env python3
"""Diagnostic: do two workgroups' LDS allocations overlap when the kernel's LDS size is not a multiple of some granule?
Workgroups of one canary kernel (tools/diag/lds_canary.hip) share CUs; each keeps writing its own pattern into its whole
allocation and then checks it (build: see tools/lds_neighbour_check.py).  For every size tried: number of words that came back with foreign content and their range."""
import ctypes, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "tgsr_amd", "lib", "diag", "liblds_canary.so"))
L.lds_canary_words.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for words in (4096, 4096 + 32, 4096 + 64, 4096 + 128, 4096 + 192, 4096 + 256, 4096 + 320, 4096 + 384, 4096 + 512, 4096 + 768,
              4096 + 1024, 4146, 12288, 12544, 12644, 15360, 15616, 13568, 10240, 8192, 20480, 20224, 40960, 13312, 13824):
    rep = torch.zeros(4 + 4 * 64, dtype=torch.int32, device="cuda")
    rep[1] = 0x7fffffff
    for _ in range(5):
        rc = L.lds_canary_words(4096, words, 2000, 1, rep.data_ptr(), st)
        assert rc == 0, (rc, words)
    torch.cuda.synchronize()
    r = rep.cpu().numpy().astype("uint32")
    msg = "LDS %6d bytes (%.3f KB): %6d foreign words" % (words * 4, words / 256.0, r[0])
    if r[0]:
        msg += "; byte range [%d, %d] of the allocation; e.g. (block, word, got, want) %s" % (
            r[1] * 4, r[2] * 4 + 3, [(int(r[4 + 4 * k]), int(r[5 + 4 * k]), hex(int(r[6 + 4 * k])), hex(int(r[7 + 4 * k]))) for k in range(2)])
    print(msg, flush=True)
