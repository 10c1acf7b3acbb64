#!/usr/bin/env python3
"""In-kernel timeline of the vision window-attention kernel (workgroup 0).  Needs the DEVELOPER build
(make -C infinitevl_amd/csrc trace).  usage: trace_vision.py [frames=1] [segment=1024] [rope 0|1]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinitevl_amd import _lib, ops
lib = _lib.load(os.path.join(ROOT, "tools", "libivl_hip_trace.so"))
dev = torch.device("cuda", 0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rope = int(sys.argv[3]) if len(sys.argv) > 3 else 1
H, d, per = 16, 80, 1024
g_ = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(frames * per, 3, H, d, device=dev, generator=g_).to(torch.bfloat16)
cs = tuple(torch.randn(frames * per, d, device=dev, generator=g_) for _ in range(2)) if rope else None
cu = torch.arange(0, frames * per + 1, seg, dtype=torch.int32, device=dev)
run = lambda: ops.vision_window_attention(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, seg, rope=cs)
trace = torch.zeros(64, dtype=torch.int64, device=dev)
lib.ivl_debug_set_trace.argtypes = [ctypes.c_void_p]
for _ in range(5):
    run()
torch.cuda.synchronize()
lib.ivl_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
for it in range(3):
    trace.zero_()
    run()
    torch.cuda.synchronize()
    t = trace.cpu().tolist()
    n = max(t[7], 1)
    print(f"--- iter {it}: tiles {t[7]} | prologue {t[0]} | per tile: barrier {t[1]//n} store+barrier {t[2]//n} QK+next loads {t[3]//n} "
          f"softmax {t[4]//n} PV {t[5]//n} (sum {sum(t[1:6])//n}) | total {t[6]}")
lib.ivl_debug_set_trace(None)
