#!/usr/bin/env python3
"""In-kernel timeline of the SWA prefill kernel (hardware block 0): shader-clock differences kept in registers and
written once at kernel end.  Needs the DEVELOPER build (make -C infinitevl_amd/csrc trace).
usage: trace_swa.py [T=256] [window=4096] [cache 0|1] [libpath]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinitevl_amd import _lib, ops
lib = _lib.load(sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "tools", "libivl_hip_trace.so"))
dev = torch.device("cuda", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cache = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B, Hq, Hkv, d = 1, 16, 2, 128
g_ = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g_).to(torch.bfloat16)
q, kn, vn = rn(B, T, Hq, d), rn(B, T, Hkv, d), rn(B, T, Hkv, d)
kw = {}
if cache:
    kw = dict(k_cache=rn(B, Hkv, W, d), v_cache=rn(B, Hkv, W, d),
              pos_dev=torch.full((1,), 10 * W, dtype=torch.int64, device=dev))
if os.environ.get("IVL_TRACE_ROPE"):           # fused M-RoPE (un-rotated q / k + tables)
    kw["rope"] = (rn(3, B, T, d), rn(3, B, T, d), [16, 24, 24])
run = lambda: ops.swa_forward(q, kn, vn, window=W, scaling=d ** -0.5, **kw)
trace = torch.zeros(64, dtype=torch.int64, device=dev)
lib.ivl_debug_set_trace.argtypes = [ctypes.c_void_p]
for it in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f"T={T} W={W} cache={cache}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
lib.ivl_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
for it in range(3):
    trace.zero_()
    trace[45] = 2 ** 62
    trace[29] = 2 ** 62
    run()
    torch.cuda.synchronize()
    t = trace.cpu().tolist()
    n = max(t[40], 1)
    if t[42] == 2:  # round-5 prefill kernel (8 compute waves, no loaders): wave 0 = key half 0, wave 4 = key half 1 (rotated)
        print(f"--- iter {it} (prefill kernel, round 5): tiles {t[40]} | prologue {t[32]} | tile loop {t[47]} = {t[47]//n} per tile | epilogue {t[38]} | total {t[39]}")
        print(f"      key half 0 per tile: QK + row max {t[33]//n} | stage tile t+2 {t[34]//n} | exps + PV {t[35]//n} | vmcnt + barrier {t[36]//n}")
        print(f"      key half 1 per tile: row max {t[50]//n} | stage tile t+2 {t[51]//n} | exps + PV + QK(t+1) {t[52]//n} | vmcnt + barrier {t[53]//n}")
        continue
    if t[42]:      # the 128-row prefill kernel (wave 0 = key half 0, wave 4 = key half 1)
        print(f"--- iter {it} (prefill kernel): tiles {t[40]} | prologue {t[32]} | tile loop {t[47]} = {t[47]//n} per tile | epilogue {t[38]} | total {t[39]} "
              f"| prologue parts: setup {t[28]}, q + rope loads issued {t[26]}, rope math {t[27]}")
        print(f"      key half 0 per tile: A (QK + row max) {t[33]//n} + wait {t[34]//n} | B (exps + PV) {t[35]//n} + wait {t[36]//n}")
        print(f"      key half 1 per tile: A (QK + row max) {t[50]//n} + wait {t[51]//n} | B (exps + PV) {t[52]//n} + wait {t[53]//n}")
        print(f"      loader wave 8 (K) per tile: issue DMA {t[55]//n} + wait {t[56]//n} | vmcnt {t[59]//n} + wait {t[60]//n}")
        print(f"      loader wave 10 (V) per tile: vmcnt {t[57]//n} + wait {t[58]//n} | issue DMA {t[62]//n} + wait {t[63]//n}")
        print(f"      start-up, loader wave 8 (K): setup {t[41]} | tiles 0, 1 issued and landed: {t[43]} || wave 10 (V) at the start-up barrier: {t[49]}")
        print(f"      launch on the 100 MHz clock: first workgroup start -> last start {(t[31]-t[29]) * 10} ns -> last end {(t[30]-t[29]) * 10} ns")
        dec = lambda v: f"{v >> 24} cycles ({(v >> 16) & 255} tiles, split {(v >> 8) & 255}, q-tile {v & 255})"
        print(f"      slowest workgroup: {dec(t[44])} | fastest: {dec(t[45])}")
        continue
    print(f"--- iter {it}: tiles {t[40]} | prologue {t[32]} | per tile: barrier1 {t[33]//n} store+barrier2 {t[34]//n} "
          f"QK issue+next loads {t[35]//n} softmax {t[36]//n} PV {t[37]//n} (sum {sum(t[33:38])//n}) | epilogue {t[38]} | total {t[39]} "
          f"| realtime ticks {t[41]} -> {t[39] / max(t[41], 1) * 100:.0f} MHz if the tick is 100 MHz")
lib.ivl_debug_set_trace(None)
