#!/usr/bin/env python3
"""Developer A/B: the fused GDN call at T tokens as two launches / one launch with the scan's workgroup width forced at
compile time (32- / 64-column workgroups): per-kernel durations from profiler records.  usage: ab_ncw.py [T=4096] [B=1] [ncw=0|2|4]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinitevl_amd import _lib, ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# product-flag builds with the width forced at compile time (ab/libivl_ncw{2,4}.so: -DIVL_AB_NCW=n), one process per build
NCW = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = _lib.load(os.path.join(ROOT, "ab", f"libivl_ncw{NCW}.so") if NCW else None)
dev = torch.device("cuda", 0)
H, K, V = 16, 128, 256
Dq, Dk, Dv = H * K, H * K, H * V
g_ = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g_).to(torch.bfloat16)
cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
cw = [rn(D_, 1, 4) * 0.5 for D_ in (Dq, Dk, Dv)]
A32, dt32 = torch.randn(H, device=dev, generator=g_), torch.randn(H, device=dev, generator=g_)
proj = rn(B, T, cols[4] + H)
cs = [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)]
h0 = (torch.randn(B, H, K, V, device=dev, generator=g_) * 0.1).to(torch.bfloat16)
from torch.profiler import ProfilerActivity, profile
def run():
    so = [c.clone() for c in cs]
    ht = torch.zeros_like(h0)
    return ops.gdn_chunk_fused(proj, cols, cw, so, so, A32, dt32, H, K, V, initial_state=h0, final_state_out=ht)
for single in (False, True):
    for ncw in (NCW,):
        ops._GDN_SINGLE_LAUNCH = single
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(10):
                run()
            torch.cuda.synchronize()
        acc = {}
        for e in prof.events():
            if "gdn_chunk" in e.name:
                nm = e.name.split("(")[0].replace("void ivl::", "")
                a = acc.setdefault(nm, [0.0, 0])
                a[0] += float(e.device_time if hasattr(e, "device_time") else e.cuda_time); a[1] += 1
        print(f"T={T} B={B} single={single} ncw={ncw}: " + "; ".join(f"{k}: {v[0] / v[1]:.1f} us x{v[1] // 10}" for k, v in acc.items()), flush=True)
ops._GDN_SINGLE_LAUNCH = True
