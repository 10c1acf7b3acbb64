#!/usr/bin/env python3
"""In-kernel timeline of the GDN chunk kernels (block 0): shader-clock stamps at phase boundaries."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import infinitevl_amd
from infinitevl_amd import ops
lib = infinitevl_amd.load_library()
dev = torch.device("cuda", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, H, K, V = 1, 16, 128, 256
g_ = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g_).to(torch.bfloat16)
q, k, v = rn(B, T, H, K), rn(B, T, H, K), rn(B, T, H, V)
beta = torch.rand(B, T, H, device=dev, generator=g_).to(torch.bfloat16)
g = torch.nn.functional.logsigmoid(torch.randn(B, T, H, device=dev, generator=g_))
state = torch.randn(B, H, K, V, device=dev, generator=g_).to(torch.bfloat16)
trace = torch.zeros(64, dtype=torch.int64, device=dev)
lib.ivl_debug_set_trace.argtypes = [ctypes.c_void_p]
for it in range(5):
    ops.chunk_gated_delta_rule(q, k, v, g, beta, initial_state=state, use_qk_l2norm_in_kernel=True, final_state_out=state)
torch.cuda.synchronize()
lib.ivl_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
for it in range(3):
    trace.zero_()
    ops.chunk_gated_delta_rule(q, k, v, g, beta, initial_state=state, use_qk_l2norm_in_kernel=True, final_state_out=state)
    torch.cuda.synchronize()
    t = trace.cpu().tolist()
    names = {0: "start", 1: "S0 done", 2: "S1 done", 3: "S2 done", 4: "S3 done", 5: "S4 done", 6: "S5 done", 7: "S6 done/end"}
    print(f"--- iter {it}: prepare kernel (cycles since kernel start)")
    for i in range(1, 8):
        print(f"   {names[i]:14s} +{t[i]-t[i-1]:8d}   (cum {t[i]-t[0]})")
    print(f"   [S0 detail] wave0 g/beta+cumsum done +{t[8]-t[0]}  zeroT done +{t[9]-t[0]}  first k data +{t[10]-t[0]}  S1a done(before barrier) +{t[11]-t[0]}")
    print("    scan kernel")
    print(f"   entry->frags0 issue     {t[17]-t[16]:8d}")
    for ci in range(min(4, (T + 63) // 64)):
        b, e = 18 + 4 * ci, 40 + 4 * ci
        prev = t[17] if ci == 0 else t[b - 1]
        print(f"   chunk {ci}: gap {t[b]-prev:6d} | publish+dma_wait {t[e]-t[b]:6d} | barrier1 {t[b+1]-t[e]:6d} | "
              f"flush+dma issue+mfma(ii) {t[e+1]-t[b+1]:6d} | vnew/eg {t[e+2]-t[e+1]:6d} | barrier2 {t[b+2]-t[e+2]:6d} | "
              f"update+out(iii) {t[b+3]-t[b+2]:6d}")
    print(f"   loop end -> stored       {t[35]-t[34]:8d}   total scan {t[35]-t[16]}   prepare->scan gap {t[16]-t[7]}")
lib.ivl_debug_set_trace(None)
