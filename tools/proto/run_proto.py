#!/usr/bin/env python3
"""Times tools/proto/libattn_proto.so (compute-loop ceiling of the one-wave-per-SIMD attention structure)."""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libattn_proto.so"))
lib.attn_proto_launch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
rows, ntiles, d = 256 * 256, 64, 128
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(rows, d, device=dev, generator=g).to(torch.bfloat16)
k = torch.randn(64, d, device=dev, generator=g).to(torch.bfloat16)
v = torch.randn(64, d, device=dev, generator=g).to(torch.bfloat16)
o = torch.empty(rows, d, device=dev, dtype=torch.bfloat16)
sc = d ** -0.5 * 1.4426950408889634
st = torch.cuda.current_stream().cuda_stream
for pipe in (0, 1, 2, 3, 4):
    for _ in range(3):
        lib.attn_proto_launch(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), rows, ntiles, sc, pipe, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        lib.attn_proto_launch(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), rows, ntiles, sc, pipe, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    flops = 4.0 * d * rows * 64 * ntiles
    # reference: softmax(q k^T / sqrt(d)) v over the one tile repeated ntiles times == over the tile once
    ref = torch.softmax((q[:512].float() @ k.float().t()) * d ** -0.5, -1) @ v.float()
    err = ((o[:512].float() - ref).norm() / ref.norm()).item()
    print(f"pipe={pipe}: {us:8.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  frac {flops / us / 1e6 / 2500:.3f}  cycles/tile/wave ~{us * 2.4e3 / ntiles:.0f}  rel err {err:.2e}")
