#!/usr/bin/env python3
"""Times tools/proto/libattn8_proto.so: the 8-wave x 32-row x 64-key attention tile loop (cdna_hip_programming.md Appendix B),
MODE 0 = tile resident in LDS (compute ceiling), 1 = every tile re-fetched by LDS-DMA from the compute waves, 2 = + QK^T of the
next tile in front of the softmax.  Builds the library when it is missing (hipcc, product flags)."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libattn8_proto.so")
src = os.path.join(HERE, "attn8_proto.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form",
                    src, "-o", so], check=True)
lib = ctypes.CDLL(so)
lib.attn8_proto_launch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
rows, d = 256 * 256, 128
ntiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
modes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 3, 4]
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(rows, d, device=dev, generator=g).to(torch.bfloat16)
k = torch.randn(64, d, device=dev, generator=g).to(torch.bfloat16)
v = torch.randn(64, d, device=dev, generator=g).to(torch.bfloat16)
o = torch.empty(rows, d, device=dev, dtype=torch.bfloat16)
sc = d ** -0.5 * 1.4426950408889634
st = torch.cuda.current_stream().cuda_stream
ref = torch.softmax((q[:512].float() @ k.float().t()) * d ** -0.5, -1) @ v.float()
# modes >= 8: REAL streams -- 2 kv heads x (ntiles x 64) keys, workgroup b reads kv head (b & 7) >> 2, a different tile per iteration
S = ntiles * 64
kr = torch.randn(2, S, d, device=dev, generator=g).to(torch.bfloat16)
vr = torch.randn(2, S, d, device=dev, generator=g).to(torch.bfloat16)


def ref_rows(wg):
    kvh = (wg & 7) >> 2
    qq = q[wg * 128:(wg + 1) * 128].float()
    return torch.softmax((qq @ kr[kvh].float().t()) * d ** -0.5, -1) @ vr[kvh].float()


for mode in modes:
    o.zero_()
    real = mode >= 8
    kk, vv = (kr, vr) if real else (k, v)
    for _ in range(3):
        lib.attn8_proto_launch(q.data_ptr(), kk.data_ptr(), vv.data_ptr(), o.data_ptr(), rows, ntiles, sc, mode, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        lib.attn8_proto_launch(q.data_ptr(), kk.data_ptr(), vv.data_ptr(), o.data_ptr(), rows, ntiles, sc, mode, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    flops = 4.0 * d * rows * 64 * ntiles
    if real:
        errs = []
        for wg in (0, 5, 300, 511):
            r_ = ref_rows(wg)
            errs.append(((o[wg * 128:(wg + 1) * 128].float() - r_).norm() / r_.norm()).item())
        err, errl = max(errs), min(errs)
    else:
        err = ((o[:512].float() - ref).norm() / ref.norm()).item()
        errl = ((o[-512:].float() - (torch.softmax((q[-512:].float() @ k.float().t()) * d ** -0.5, -1) @ v.float())).norm() / ref.norm()).item()
    print(f"mode={mode}: {us:8.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  frac {flops / us / 1e6 / 2500:.3f}  cycles/tile @2.4GHz ~{us * 2.4e3 / ntiles:.0f}  "
          f"rel err {err:.2e} / {errl:.2e}", flush=True)
