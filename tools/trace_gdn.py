#!/usr/bin/env python3
"""In-kernel timeline of the GDN chunk kernels (block 0): shader-clock differences kept in registers and written once
at kernel end.  Needs the DEVELOPER build (make -C infinitevl_amd/csrc trace); usage: trace_gdn.py [T] [scan waves 2|4]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinitevl_amd import _lib, ops
lib = _lib.load(sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "tools", "libivl_hip_trace.so"))
dev = torch.device("cuda", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib.ivl_debug_set_scan_waves(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
B, H, K, V = 1, 16, 128, 256
g_ = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g_).to(torch.bfloat16)
q, k, v = rn(B, T, H, K), rn(B, T, H, K), rn(B, T, H, V)
beta = torch.rand(B, T, H, device=dev, generator=g_).to(torch.bfloat16)
g = torch.nn.functional.logsigmoid(torch.randn(B, T, H, device=dev, generator=g_))
state = torch.randn(B, H, K, V, device=dev, generator=g_).to(torch.bfloat16)
trace = torch.zeros(128, dtype=torch.int64, device=dev)
lib.ivl_debug_set_trace.argtypes = [ctypes.c_void_p]
run = lambda: ops.chunk_gated_delta_rule(q, k, v, g, beta, initial_state=state, use_qk_l2norm_in_kernel=True,
                                         final_state_out=state)
if os.environ.get("IVL_TRACE_TWO"):            # the two-launch form of the fused call
    ops._GDN_SINGLE_LAUNCH = False
if os.environ.get("IVL_TRACE_FUSED"):          # the pre-pass with the conv / gate front end (developer tool only)
    Dq, Dk, Dv = H * K, H * K, H * V
    cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
    proj = rn(B, T, cols[4] + H)
    cw = [rn(D_, 1, 4) for D_ in (Dq, Dk, Dv)]
    cs = [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)]
    A32, dt32 = torch.randn(H, device=dev, generator=g_), torch.randn(H, device=dev, generator=g_)
    run = lambda: ops.gdn_chunk_fused(proj, cols, cw, cs, cs, A32, dt32, H, K, V, initial_state=state, final_state_out=state)
if os.environ.get("IVL_TRACE_COLD"):           # every traced launch behind a 1 GiB read sweep: its inputs come from HBM (as inside a real call)
    big_ = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    run_hot = run
    run = lambda: (big_.sum(), run_hot())[1]
for it in range(5):
    run()
torch.cuda.synchronize()
lib.ivl_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
NT = (T + 63) // 64
for it in range(3):
    trace.zero_()
    trace[26] = 2 ** 62
    run()
    torch.cuda.synchronize()
    t = trace.cpu().tolist()
    print(f"--- iter {it}: prepare (cycles): load+l2norm {t[1]} | L,A mfma {t[2]} | solve L0 {t[3]} L1 {t[4]} L2 {t[5]} | Tu,w {t[6]} "
          f"| total {t[7]-t[0]}")
    print(f"    fused front end (wave 0): loads issued {t[8]} | state hand-over {t[9]} | conv q {t[10]} | conv k {t[11]}")
    print(f"    v wave 7: loads issued {t[12]} | state hand-over {t[13]} | conv v {t[14]} | arrival at B1: wave 0 {t[15]}, wave 1 {t[30]}, wave 7 {t[29]}")
    print(f"    prepare launch on the 100 MHz clock: first workgroup start -> last end {(t[27]-t[26]) * 10} ns, longest workgroup {t[28] * 10} ns")
    if t[42]:
        print(f"    single launch, ns after the first pre-pass workgroup's start: scan workgroup 0 starts {(t[41]-t[26]) * 10} | last pre-pass end {(t[27]-t[26]) * 10} "
              f"| flags seen {(t[42]-t[26]) * 10} | PA {(t[43]-t[26]) * 10} | scan end {(t[44]-t[26]) * 10}")
        print(f"      pre-pass workgroup 0 (k side of chunk 0), cycles: conv-state wait + write {t[49]} | store drain {t[50]} | barrier {t[51]}")
        print(f"      k side: last start {(t[48]-t[26]) * 10} last publish {(t[46]-t[26]) * 10} | q side: last start {(t[47]-t[26]) * 10 if t[47] else None} last publish {(t[45]-t[26]) * 10 if t[45] else None}")
    print(f"    scan: total {t[22]-t[16]} | loop {t[20]} (per chunk: sb cvt+publish {t[24]//NT} | T wait {t[17]//NT} | phase A + vn publish {t[18]//NT} | M wait {t[23]//NT} | phase B {t[19]//NT}) "
          f"| state store {t[21]} | prepare end -> scan start {t[16]-t[7]} "
          f"| scan realtime ticks {t[25]} -> {(t[22]-t[16]) / max(t[25], 1) * 100:.0f} MHz if the tick is 100 MHz")
    for L in range(4):
        o = 64 + 8 * L
        print(f"    loader {L}: vmcnt wait before T {t[o]//NT} | T wait {t[o+1]//NT} | issue H2/VT {t[o+2]//NT} | vmcnt wait before M {t[o+3]//NT} | M wait {t[o+4]//NT} | issue H1/touch {t[o+5]//NT}")
    print("    waits at T | M per chunk -- state waves:", [(t[120+2*i]//NT, t[121+2*i]//NT) for i in range(4)], " output waves:",
          [(t[112+2*i]//NT, t[113+2*i]//NT) for i in range(4)], " V waves:", [(t[104+2*i]//NT, t[105+2*i]//NT) for i in range(4)])
    print(f"    output wave 0: T wait {t[100]//NT} | phase A {t[101]//NT} | M wait {t[102]//NT} | phase B {t[103]//NT}")
    print(f"    V wave 0: conv(0) {t[31]} | P0 wait + mma(0) {t[32]} | P wait {t[33]} | per chunk: T wait {t[34]//NT} | mma {t[35]//NT} | M wait {t[36]//NT} | conv {t[37]//NT} || conv-only {t[38]//NT} take {t[39]//NT} issue {t[40]//NT}")
lib.ivl_debug_set_trace(None)
