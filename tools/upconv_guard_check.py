#!/usr/bin/env python3
"""Diagnostic: does lp_upconv_glu_kernel write outside its output image?  The image sits inside one allocation between two
guard bands holding a pattern; so do the input image and (HK > 0) the partial-sum buffer."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tgsr_amd import lp
dev = "cuda"
g = torch.Generator().manual_seed(0)
R = lambda *sh: torch.randn(*sh, generator=g).to(dev)
GUARD = 1 << 20     # elements of bf16 on either side
for (cin, hw, K) in ((32, 32, 0), (32, 32, 5), (64, 64, 3), (32, 64, 0), (32, 128, 5)):
    B = 16
    n_out = B * (2 * hw + 2) * (2 * hw + 2) * 32
    big = torch.full((GUARD + n_out + GUARD,), 7.0, dtype=torch.bfloat16, device=dev)
    out = big[GUARD:GUARD + n_out].view(B, 2 * hw + 2, 2 * hw + 2, 32)
    out.zero_()
    n_in = B * (hw + 2) * (hw + 2) * cin
    bigx = torch.full((GUARD + n_in + GUARD,), 5.0, dtype=torch.bfloat16, device=dev)
    x = bigx[GUARD:GUARD + n_in].view(B, hw + 2, hw + 2, cin)
    x.copy_(lp.from_nchw(R(B, cin, hw, hw), "bf16"))
    wp = lp.pack_upconv_weight(R(64, cin, 3, 3) * 0.1, "bf16")
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    if K:
        hwp = lp.pack_to3_weight(R(3, 32, K, K) * 0.1, "bf16")
        npart = lp.head_partial_elems(B, 2 * hw, 2 * hw, K)
        bigp = torch.full((GUARD + npart + GUARD,), 3.0, dtype=torch.float32, device=dev)
        part = bigp[GUARD:GUARD + npart]
    for _ in range(5):
        if K:
            lp.upconv_glu_head(x, wp, cin, 64, sc, sh, hwp, K, partial=part, out=out)
        else:
            lp.upconv_glu(x, wp, cin, 64, sc, sh, out=out)
    torch.cuda.synchronize()
    msg = "upconv cin %d @%d->%d K %d:" % (cin, hw, 2 * hw, K)
    for name, bb, gval, n in (("out", big, 7.0, n_out), ("x", bigx, 5.0, n_in)) + ((("partial", bigp, 3.0, npart),) if K else ()):
        lo, hi = bb[:GUARD], bb[GUARD + n:]
        nb = int((lo != gval).sum()) + int((hi != gval).sum())
        msg += "  %s guards: %d changed" % (name, nb)
        if nb:
            il = (lo != gval).nonzero().flatten(); ih = (hi != gval).nonzero().flatten()
            msg += " (below: %s, above: %s)" % ((GUARD - il[:4]).tolist() if len(il) else [], ih[:4].tolist() if len(ih) else [])
    # border of the image itself must stay zero
    o = out.float()
    border = float(o[:, 0].abs().sum() + o[:, -1].abs().sum() + o[:, :, 0].abs().sum() + o[:, :, -1].abs().sum())
    msg += "  ; image border sum %.3f" % border
    print(msg, flush=True)
