#!/usr/bin/env python3
"""In-kernel timeline of the SWA prefill kernel (hardware block 0): shader-clock stamps per tile phase."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import infinitevl_amd
from infinitevl_amd import ops
lib = infinitevl_amd.load_library()
dev = torch.device("cuda", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = 4096 if T <= 4096 else 2 * T
B, Hq, Hkv, d = 1, 16, 2, 128
g_ = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g_).to(torch.bfloat16)
C = W - 1
kc, vc = rn(B, Hkv, C, d), rn(B, Hkv, C, d)
pos_dev = torch.full((1,), 10 * W, dtype=torch.int64, device=dev)
q, kn, vn = rn(B, T, Hq, d), rn(B, T, Hkv, d), rn(B, T, Hkv, d)
trace = torch.zeros(64, dtype=torch.int64, device=dev)
lib.ivl_debug_set_trace.argtypes = [ctypes.c_void_p]
run = lambda: ops.swa_forward(q, kn, vn, window=W, scaling=d ** -0.5, k_cache=kc, v_cache=vc, pos_dev=pos_dev)
for _ in range(5):
    run()
torch.cuda.synchronize()
lib.ivl_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
for it in range(2):
    trace.zero_(); run(); torch.cuda.synchronize()
    t = trace.cpu().tolist()
    print(f"--- iter {it}: prologue(Q frags etc) -> first tile loop: {t[1]-t[0]}")
    for i in range(6):
        b = 1 + 5 * i
        if t[b] == 0: break
        nxt = t[b + 5] if (i < 5 and t[b + 5]) else t[40]
        print(f"  tile {i}: stage(2 barriers) {t[b+1]-t[b]:6d}  QK^T {t[b+2]-t[b+1]:6d}  softmax {t[b+3]-t[b+2]:6d}  PV(+loop) {nxt-t[b+3]:6d}   total {nxt-t[b]:6d}")
lib.ivl_debug_set_trace(None)
