#!/usr/bin/env python3
"""Developer check: the single-launch form of the fused GDN call against the two-launch form, bit for bit, many times.
usage: check_single.py [libpath] [T=256] [H=16] [iters=40]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinitevl_amd import _lib, ops
if len(sys.argv) > 1 and sys.argv[1]:
    _lib.load(sys.argv[1])
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = int(sys.argv[3]) if len(sys.argv) > 3 else 16
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 40
dev = torch.device("cuda", 0)
B, K, V = 1, 128, 256
Dq, Dk, Dv = H * K, H * K, H * V
g_ = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g_).to(torch.bfloat16)
cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
cw = [rn(D_, 1, 4) * 0.5 for D_ in (Dq, Dk, Dv)]
A32, dt32 = torch.randn(H, device=dev, generator=g_), torch.randn(H, device=dev, generator=g_)
# phase 1: the two-launch reference of every input set; phase 2: the single-launch form on the same sets, back to back -- a
# record line left in some L2 by an earlier launch would then belong to DIFFERENT inputs
sets, refs = [], []
def run(st, single):
    proj, cs, h0 = st
    ops._GDN_SINGLE_LAUNCH = single
    so = [c.clone() for c in cs]
    ht = torch.zeros_like(h0)
    o = ops.gdn_chunk_fused(proj, cols, cw, so, so, A32, dt32, H, K, V, initial_state=h0, final_state_out=ht)
    ops._GDN_SINGLE_LAUNCH = True
    return o, ht, so
for it in range(iters):
    st = (rn(B, T, cols[4] + H), [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)], (torch.randn(B, H, K, V, device=dev, generator=g_) * 0.1).to(torch.bfloat16))
    sets.append(st)
    refs.append(run(st, False))
torch.cuda.synchronize()
bad = 0
for rep in range(3):
    outs = [run(st, True) for st in sets]
    torch.cuda.synchronize()
    for it, (a_, b_) in enumerate(zip(refs, outs)):
        ok = torch.equal(a_[0], b_[0]) and torch.equal(a_[1], b_[1]) and all(torch.equal(x, y) for x, y in zip(a_[2], b_[2]))
        if not ok:
            bad += 1
            d = (a_[0].float() - b_[0].float()).abs()
            nz = (d > 0).nonzero()
            print(f"rep {rep} set {it}: MISMATCH o: {int((d > 0).sum())} elements, max {float(d.max()):.3e}; first at {nz[0].tolist() if len(nz) else None}")
area = ops._gdn_sync_area(dev)
print(f"T={T} H={H}: {bad} of {3 * iters} single-launch calls differ from the two-launch form; sync area sum {int(area.view(torch.int32).abs().sum())}")
