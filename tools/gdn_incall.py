#!/usr/bin/env python3
"""Where do the ~20 us go between the hot micro-benchmark of the long-call GDN launch (T = 4096: 80-87 us) and the same launch
inside a configs[3] call (100 us)?  (VERDICT r4 #4.)  The launch is timed (profiler activity records) behind four predecessors:
  hot        : the previous launch of the same call (its inputs are in the L2s / the Infinity Cache)
  evicted    : a read sweep of 1 GiB in front (inputs come from HBM; nothing of the inputs is dirty)
  rewritten  : the projection buffer (101 MB) rewritten by a copy kernel in front (what the in-projection GEMM does: the 101 MB
               are dirty in the writing XCDs' L2s at the kernel boundary and come back from the memory side)
  gemm       : the projection written by the real [4096 x 2048] x [2048 x 12320] library GEMM in front
  gemm+idle  : the same GEMM, then ~40 us of an idle spin kernel (torch.cuda._sleep) before the launch: is it the GEMM's write-back
               still draining when the launch starts (then this equals `evicted`), or where its output ends up?
  gemm+sweep : the GEMM, then a 256 MB read sweep (the output leaves the L2s / the Infinity Cache clean before the launch)
usage: gdn_incall.py [T=4096] [libpath|-] [only=<hot|evicted|rewritten|gemm>]
With `only` (the counter passes of tools/pmc.sh: one predecessor per process, no torch.profiler beside rocprofv3) the launch pair is
just repeated 12 times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from infinitevl_amd import _lib, ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if len(sys.argv) > 2 and sys.argv[2] != "-":
    _lib.load(sys.argv[2])
ONLY = sys.argv[3] if len(sys.argv) > 3 else None
dev = torch.device("cuda", 0)
B, H, K, V = 1, 16, 128, 256
Dq, Dk, Dv = H * K, H * K, H * V
g_ = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g_).to(torch.bfloat16)   # noqa: E731
cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
ld = cols[4] + H
cw = [rn(D_, 1, 4) * 0.5 for D_ in (Dq, Dk, Dv)]
A32, dt32 = torch.randn(H, device=dev, generator=g_), torch.randn(H, device=dev, generator=g_)
proj, src = rn(B, T, ld), rn(B, T, ld)
cs = [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)]
state = (torch.randn(B, H, K, V, device=dev, generator=g_) * 0.1).to(torch.bfloat16)
x = rn(B * T, 2048) * 0.05
w = rn(ld, 2048) * 0.05
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)          # 1 GiB


def gdn():
    return ops.gdn_chunk_fused(proj, cols, cw, cs, cs, A32, dt32, H, K, V, initial_state=state, final_state_out=state)


def pre_hot():
    pass


def pre_evict():
    big.sum()


def pre_rewrite():
    proj.copy_(src)


def pre_gemm():
    torch.mm(x, w.t(), out=proj.view(B * T, ld))


def pre_gemm_idle():
    pre_gemm()
    torch.cuda._sleep(80000)


small = torch.empty(1 << 26, dtype=torch.float32, device=dev)        # 256 MB


def pre_gemm_sweep():
    pre_gemm()
    small.sum()


for name, pre in (("hot", pre_hot), ("evicted", pre_evict), ("rewritten", pre_rewrite), ("gemm", pre_gemm), ("gemm+idle", pre_gemm_idle),
                  ("gemm+sweep", pre_gemm_sweep)):
    if ONLY is not None:
        if name == ONLY:
            for _ in range(12):
                pre(); gdn()
            torch.cuda.synchronize()
        continue
    for _ in range(3):
        pre(); gdn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            pre(); gdn()
        torch.cuda.synchronize()
    acc = {}
    for e in prof.events():
        if "gdn_chunk" in e.name:
            nm = e.name.split("(")[0].replace("void ivl::", "")
            a = acc.setdefault(nm, [0.0, 0])
            a[0] += float(e.device_time if hasattr(e, "device_time") else e.cuda_time); a[1] += 1
    print(f"T={T} predecessor={name:10s}: " + "; ".join(f"{k}: {v[0] / v[1]:.1f} us x{v[1]}" for k, v in acc.items()), flush=True)
