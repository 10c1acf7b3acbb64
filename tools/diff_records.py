#!/usr/bin/env python3
"""Developer check: which region of the pre-pass record differs between the two-launch and the single-launch form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinitevl_amd import _lib, ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
B, K, V = 1, 128, 256
Dq, Dk, Dv = H * K, H * K, H * V
g_ = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g_).to(torch.bfloat16)
cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
cw = [rn(D_, 1, 4) * 0.5 for D_ in (Dq, Dk, Dv)]
A32, dt32 = torch.randn(H, device=dev, generator=g_), torch.randn(H, device=dev, generator=g_)
proj = rn(B, T, cols[4] + H)
cs = [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)]
h0 = (torch.randn(B, H, K, V, device=dev, generator=g_) * 0.1).to(torch.bfloat16)
recs = []
for single in (False, True):
    ops._GDN_SINGLE_LAUNCH = single
    so = [c.clone() for c in cs]
    ht = torch.zeros_like(h0)
    o = ops.gdn_chunk_fused(proj, cols, cw, so, so, A32, dt32, H, K, V, initial_state=h0, final_state_out=ht)
    torch.cuda.synchronize()
    ws = ops._gdn_workspace(1, ops._gdn_sync_area(dev))
    NT = (T + 63) // 64
    recs.append(ws[: B * H * NT * 62464].clone().view(B * H, NT, 62464).cpu())
regions = dict(WN=(0, 16384), QH=(16384, 32768), KDT=(32768, 49152), AQK=(49152, 55296), EG=(55296, 55552), EGL=(55552, 55556),
               BETA=(55808, 55936), TU=(56320, 62464))
a, b = recs
for name, (lo, hi) in regions.items():
    d = (a[:, :, lo:hi] != b[:, :, lo:hi])
    print(f"{name:5s}: {int(d.sum())} differing bytes of {d.numel()}", end="")
    if d.any():
        idx = d.nonzero()
        print(f"  first (bh, chunk, byte) = {idx[0].tolist()}; chunks {sorted(set(idx[:, 1].tolist()))}; bytes {idx[:, 2].min().item()}..{idx[:, 2].max().item()}")
    else:
        print()
