#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): prefill tok/s + decode tok/s, InfiniteVL-3B @128K sequence.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                            torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): the full 36-layer InfiniteVL-3B decoder (random-init bf16 weights of
the shipped config.json -- no checkpoint offline), streaming prefill in 256-token chunk steps with the
sliding window overridden to 4096, one hipGraph replay per step, one sequence per GPU (batch-sharded
replicas; the only collective is the final logits all-gather).  A "step" = one 256-token chunk through
all 36 layers (27 Gated DeltaNet + 9 SWA mixers on the gfx950 kernels of this repo, the projections/MLP
on stock rocBLAS/hipBLASLt).  Untimed setup streams 131072 - K*256 tokens (at least the 17 steps that fill the SWA window), so that the K
timed steps are steady-state full-window steps that END at exactly 128K tokens of context whatever K is.
After the timed prefill region a decode leg (graph-captured greedy single-token steps at that context)
is timed separately and reported as `decode_tok_s`.

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (dominant hot-path kernel, measured live with
HIP events on the launch stream), `kernels` (all hot-path kernels), `cpu_baseline` (oracle on host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _enable_tunableop():
    """The dense projections / MLP are stock rocBLAS / hipBLASLt GEMMs reached through torch (out of the
    path's scope); let PyTorch's own TunableOp pick the library kernel per GEMM shape (+6 % prefill, +35 %
    decode at these M=256 / M=1 shapes).  A results file recorded on gfx950 / ROCm 7.2 is shipped and
    pre-seeded per rank; shapes it does not cover are tuned during the warm-up steps (seconds)."""
    if os.environ.get("IVL_NO_TUNABLEOP") == "1":
        return
    import shutil
    import tempfile
    rank = os.environ.get("LOCAL_RANK", "0")
    d = tempfile.mkdtemp(prefix=f"ivl_tunableop_r{rank}_")
    seed = os.path.join(ROOT, "infinitevl_amd", "tuning", "tunableop_gfx950_rocm72.csv")
    if os.path.exists(seed):
        for ordinal in range(8):
            shutil.copy(seed, os.path.join(d, f"results{ordinal}.csv"))
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(d, "results.csv"))


_enable_tunableop()
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=256, help="tokens per streaming step")
    ap.add_argument("--window", type=int, default=4096)
    ap.add_argument("--context", type=int, default=131072, help="context length at the END of the timed region")
    ap.add_argument("--decode-steps", type=int, default=128)
    ap.add_argument("--layers", type=int, default=36, help="debug only; the reported config is 36")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg1", action="store_true", help="skip the 4K one-call prefill + decode leg (configs[1])")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-fp8", action="store_true", help="skip the fp8 (e4m3) leg (BASELINE.json configs[4])")
    return ap.parse_args()


def event_time_ms(fn, iters, stream):
    """Average GPU duration of `fn`: `iters` calls are captured into one hipGraph (so the measurement is
    not bound by the Python/ctypes launch overhead of ~10-30 us per call) and the replay is timed with HIP
    events recorded on the stream the kernels run on."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record(cur)
    for _ in range(reps):
        graph.replay()
    e1.record(cur)
    e1.synchronize()
    return e0.elapsed_time(e1) / (iters * reps)


def kernel_timings(device, chunk, window, only=None):
    """Per-launch GPU time of every hot-path kernel at the bench shapes (B=1, T=chunk, InfiniteVL-3B head
    shapes, full window), plus two throughput-regime shapes (T=4096) for the two heavy kernels.  Each entry:
    ms per launch, launches per prefill step, the roofline that bounds it, algorithmic bytes/flops per
    launch and the achieved rate."""
    from infinitevl_amd import ops
    st = torch.cuda.current_stream(device)
    B, T, H, K, V, Hq, Hkv, d = 1, chunk, 16, 128, 256, 16, 2, 128
    g_ = torch.Generator(device=device).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=device, generator=g_).to(torch.bfloat16)  # noqa: E731
    C = window - 1
    res = {}

    def add(name, fn, iters, per_step, bound, work):
        if only == "!large":                       # the PMC passes: bench-shape launches only (keyed by kernel + grid)
            if "@T=" in name or "@B=" in name:
                return
        elif only is not None and only not in name:
            return
        ms = event_time_ms(fn, iters, st)
        if bound == "hbm":
            res[name] = dict(ms=ms, launches_per_step=per_step, bound="hbm", alg_bytes=work,
                             achieved=work / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        else:
            res[name] = dict(ms=ms, launches_per_step=per_step, bound="mfma", alg_flops=work,
                             achieved=work / (ms * 1e-3) / 1e12, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s")
        res[name]["frac"] = res[name]["achieved"] / res[name]["peak"]

    def gdn_inputs(Tn):
        q, k, v = rn(B, Tn, H, K), rn(B, Tn, H, K), rn(B, Tn, H, V)
        beta = torch.rand(B, Tn, H, device=device, generator=g_).to(torch.bfloat16)
        g = torch.nn.functional.logsigmoid(torch.randn(B, Tn, H, device=device, generator=g_))
        return q, k, v, g, beta

    state = torch.randn(B, H, K, V, device=device, generator=g_).to(torch.bfloat16)
    # GDN bytes: 24,672 B/token/layer + state read+write per call in the dtype actually passed (bf16 cache: SURVEY.md Q5)
    sbytes = 2 * H * K * V * state.element_size()
    q, k, v, g, beta = gdn_inputs(T)
    add("gdn_chunk(prepare+scan)", lambda: ops.chunk_gated_delta_rule(
        q, k, v, g, beta, initial_state=state, use_qk_l2norm_in_kernel=True, final_state_out=state),
        20, 0, "hbm", 24672.0 * T + sbytes)
    add("gdn_chunk_fp8(prepare+scan)", lambda: ops.chunk_gated_delta_rule(
        q, k, v, g, beta, initial_state=state, use_qk_l2norm_in_kernel=True, final_state_out=state, mma_dtype="fp8_e4m3"),
        20, 0, "hbm", 24672.0 * T + sbytes)
    # SWA prefill: T queries over a full ring (W-1 cached keys) + T new keys; 8192*min(p+1,W) FLOP/token/layer
    kc, vc = rn(B, Hkv, C, d), rn(B, Hkv, C, d)
    pos_dev = torch.full((1,), 10 * window, dtype=torch.int64, device=device)
    qs, kn, vn = rn(B, T, Hq, d), rn(B, T, Hkv, d), rn(B, T, Hkv, d)
    add("swa_prefill", lambda: ops.swa_forward(qs, kn, vn, window=window, scaling=d ** -0.5, k_cache=kc, v_cache=vc,
                                               pos_dev=pos_dev), 20, 9, "mfma", 4.0 * Hq * d * window * T)
    add("swa_cache_append", lambda: ops.swa_cache_append(kn, vn, kc, vc, pos_dev=pos_dev), 50, 9, "hbm",
        4.0 * T * Hkv * d * 2)
    # fused prologue (3 convs + gate math from one projection buffer), gated norm, decoder-layer norm
    Dq, Dk, Dv = H * K, H * K, H * V
    cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
    ld = cols[4] + H
    proj = rn(B, T, ld)
    cw = [rn(D_, 1, 4) for D_ in (Dq, Dk, Dv)]
    cs = [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)]
    A32, dt32 = torch.randn(H, device=device, generator=g_), torch.randn(H, device=device, generator=g_)
    # the step's GDN layers run the chunk kernel WITH its front end (3 convs + SiLU + gate math inside the pre-pass): the
    # projection is read once (16,416 B/token incl. the a / b gate inputs) and q / k / v / g / beta never reach HBM;
    # algorithmic bytes per token: 16,416 (projection slice) + 8,192 (o) = 24,608 B + the state
    add("gdn_chunk_fused(convs+gates+prepare+scan)", lambda: ops.gdn_chunk_fused(
        proj, cols, cw, cs, cs, A32, dt32, H, K, V, initial_state=state, final_state_out=state), 20, 27, "hbm",
        24608.0 * T + sbytes)
    add("gdn_prologue(3 convs + gates)", lambda: ops.gdn_prologue(proj, cols, cw, cs, cs, A32, dt32, H, Dq, Dk, Dv),
        50, 0, "hbm", 2.0 * T * 8192 * 2 + T * H * (2 * 2 + 4 + 2))
    xo, wn = rn(B, T, H, V), rn(V)
    add("rmsnorm_swish_gate", lambda: ops.rmsnorm_swish_gate_strided(xo, proj[..., Dq + Dk + Dv:], ld, wn, 1e-5),
        50, 27, "hbm", 3.0 * T * H * V * 2)
    xh, rh, wh = rn(B, T, 2048), rn(B, T, 2048), rn(2048)
    add("add_rmsnorm(decoder layer)", lambda: ops.add_rmsnorm(xh, rh, wh, 1e-6), 50, 72, "hbm", 4.0 * T * 2048 * 2)
    # decode-shape kernels (per decode token, not per prefill step)
    q1, k1, v1, g1, b1 = gdn_inputs(1)
    add("gdn_recurrent(decode)", lambda: ops.fused_recurrent_gated_delta_rule(
        q1, k1, v1, g1, b1, initial_state=state, use_qk_l2norm_in_kernel=True, final_state_out=state),
        100, 27, "hbm", 24672.0 + sbytes)
    proj1 = rn(B, 1, ld)
    cols6 = (cols[0], cols[1], cols[2], Dq + Dk + Dv, cols[3], cols[4])
    add("gdn_decode_step(decode: convs+gates+rule+norm, 1 launch)", lambda: ops.gdn_decode_step(
        proj1, cols6, cw, cs, A32, dt32, wn, 1e-5, state, H, K, V, K ** -0.5), 100, 0, "hbm",
        2.0 * ld + sbytes + 2.0 * H * V)
    qd, kd1, vd1 = rn(B, 1, Hq, d), rn(B, 1, Hkv, d), rn(B, 1, Hkv, d)
    add("swa_decode", lambda: ops.swa_forward(qd, kd1, vd1, window=window, scaling=d ** -0.5, k_cache=kc, v_cache=vc,
                                              pos_dev=pos_dev), 100, 9, "hbm", 1024.0 * window)
    # decode-step projections (M = 1 weight streams; outside SURVEY.md section 8's rows, listed for the decode leg)
    for nm, N_, K_, per in (("gdn in-proj", 12320, 2048, 27), ("mlp gate|up", 22016, 2048, 36),
                            ("mlp down", 2048, 11008, 36), ("gdn o_proj", 2048, 4096, 27), ("lm_head", 151936, 2048, 1)):
        w_, x_ = rn(N_, K_), rn(1, 1, K_)
        add(f"decode linear {nm} [{N_}x{K_}]", lambda w_=w_, x_=x_: ops.linear(x_, w_), 50, 0, "hbm",
            2.0 * N_ * K_ + 2.0 * (N_ + K_))
        res_key = f"decode linear {nm} [{N_}x{K_}]"
        if res_key in res:
            res[res_key]["launches_per_decode_token"] = per
        del w_, x_
    # batched streams (8 sequences per GPU in one call): the same kernels with 8x the work per launch -- shows that
    # the low fractions at B=1 come from the size of a 256-token step, not from the kernels
    Bb = 8
    qb, kb_, vb_ = rn(Bb, T, H, K), rn(Bb, T, H, K), rn(Bb, T, H, V)
    betab = torch.rand(Bb, T, H, device=device, generator=g_).to(torch.bfloat16)
    gb_ = torch.nn.functional.logsigmoid(torch.randn(Bb, T, H, device=device, generator=g_))
    stateb = torch.randn(Bb, H, K, V, device=device, generator=g_).to(torch.bfloat16)
    add("gdn_chunk@B=8", lambda: ops.chunk_gated_delta_rule(
        qb, kb_, vb_, gb_, betab, initial_state=stateb, use_qk_l2norm_in_kernel=True, final_state_out=stateb),
        10, 0, "hbm", Bb * (24672.0 * T + sbytes))
    add("gdn_chunk_fp8@B=8", lambda: ops.chunk_gated_delta_rule(
        qb, kb_, vb_, gb_, betab, initial_state=stateb, use_qk_l2norm_in_kernel=True, final_state_out=stateb,
        mma_dtype="fp8_e4m3"), 10, 0, "hbm", Bb * (24672.0 * T + sbytes))
    kcb, vcb = rn(Bb, Hkv, C, d), rn(Bb, Hkv, C, d)
    qsb, knb, vnb = rn(Bb, T, Hq, d), rn(Bb, T, Hkv, d), rn(Bb, T, Hkv, d)
    add("swa_prefill@B=8", lambda: ops.swa_forward(qsb, knb, vnb, window=window, scaling=d ** -0.5, k_cache=kcb,
                                                   v_cache=vcb, pos_dev=pos_dev), 10, 0, "mfma",
        Bb * 4.0 * Hq * d * window * T)
    del qb, kb_, vb_, kcb, vcb, qsb, knb, vnb
    # throughput-regime shapes (one-shot 4096-token prefill, BASELINE.json configs[1]); not part of the step
    TL = 4096
    qL, kL, vL, gL, bL = gdn_inputs(TL)
    add("gdn_chunk@T=4096", lambda: ops.chunk_gated_delta_rule(
        qL, kL, vL, gL, bL, initial_state=state, use_qk_l2norm_in_kernel=True, final_state_out=state),
        5, 0, "hbm", 24672.0 * TL + sbytes)
    add("gdn_chunk_fp8@T=4096", lambda: ops.chunk_gated_delta_rule(
        qL, kL, vL, gL, bL, initial_state=state, use_qk_l2norm_in_kernel=True, final_state_out=state, mma_dtype="fp8_e4m3"),
        5, 0, "hbm", 24672.0 * TL + sbytes)
    qsL, knL, vnL = rn(B, TL, Hq, d), rn(B, TL, Hkv, d), rn(B, TL, Hkv, d)
    add("swa_prefill@T=4096(causal)", lambda: ops.swa_forward(qsL, knL, vnL, window=8192, scaling=d ** -0.5),
        5, 0, "mfma", 4.0 * Hq * d * (TL * (TL + 1) / 2))
    # vision tower (not part of the text-stack step): 8 frames of 32 x 32 patches, 16 heads x 80,
    # rotary embedding folded in; a window layer (64-patch segments) and a full-attention layer (one segment per frame).
    # flops = 4 * d * H * sum(len^2) (non-causal)
    Hv, dv, frames, per = 16, 80, 8, 1024
    qkv = rn(frames * per, 3, Hv, dv)
    vcos, vsin = (torch.randn(frames * per, dv, device=device, generator=g_) for _ in range(2))
    for tag, seg in (("window layer, 64-patch segments", 64), ("full layer, 1024-patch segments", 1024)):
        cu = torch.arange(0, frames * per + 1, seg, dtype=torch.int32, device=device)
        add(f"vision_attn({tag})@8 frames", lambda cu=cu, seg=seg: ops.vision_window_attention(
            qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, seg, rope=(vcos, vsin)), 20, 0, "mfma",
            4.0 * dv * Hv * (frames * per // seg) * seg * seg)
    return res


def pmc_traffic(kernel_name, chunk, window):
    """HBM-side bytes per launch of a hot-path kernel from the committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_traffic.json, produced by tools/collect_profiles.sh on this same command's kernels at the
    default bench shapes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot
    run the PMC passes itself, so the newest committed measurement is attached; null when the shapes differ."""
    import glob
    if chunk != 256 or window != 4096:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    k = json.load(open(files[-1]))["kernels"]
    want = {
        "gdn_chunk(prepare+scan)": [("ivl::gdn_chunk_prepare_kernel<false, false>", 32768), ("ivl::gdn_chunk_scan_kernel<2, false>", 65536)],
        "gdn_chunk_fused(convs+gates+prepare+scan)": [("ivl::gdn_chunk_prepare_kernel<false, true>", 32768),
                                                      ("ivl::gdn_chunk_scan_kernel<2, false>", 65536)],
        "swa_prefill": [("ivl::swa_prefill_kernel", 196608), ("ivl::swa_combine_kernel<8, true>", 270336)],
        "gdn_prologue(3 convs + gates)": [("ivl::gdn_prologue_kernel", 69632)],
        "add_rmsnorm(decoder layer)": [("ivl::add_rmsnorm_kernel", 65536)],
        "rmsnorm_swish_gate": [("ivl::rmsnorm_gate_strided_kernel", 131072)],
        "gdn_recurrent(decode)": [("ivl::gdn_recurrent_kernel", 32768)],
    }.get(kernel_name)
    if not want:
        return None
    parts = []
    for base, grid in want:          # kernel names carry template arguments: match on the prefix and the grid size
        hit = [n for n in k if n.startswith(base) and n.endswith(f"@grid{grid}")]
        if len(hit) != 1:
            return None
        parts.append(hit[0])
    return {"hbm_bytes": sum(k[p_]["hbm_bytes"] for p_ in parts), "source": os.path.basename(files[-1]),
            "launches": parts}


def cpu_baseline(chunk, window):
    """Oracle (CPU restatement of the path = BASELINE.md section 3 'reference CPU eager path') on the host cores:
    ONE 4-layer period (1 SWA + 3 GDN decoder layers, real InfiniteVL-3B shapes, fp32 weights) for one
    `chunk`-token step with a full window, extrapolated x9 to the 36-layer stack."""
    from oracle import model as omodel
    from oracle.cache import SwaCounters
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    # small-operator torch CPU code collapses when oversubscribed (72 s/step on 256 threads vs 0.5 s on 8)
    torch.set_num_threads(max(1, min(ncores, 32)))
    oc = omodel.OracleConfig(sliding_window=window, layer_types=["sliding_attention"] + ["linear_attention"] * 3)
    params = omodel.random_params(oc, seed=0)
    cache = omodel.new_cache(oc)
    # full window without paying for the fill: plant W-1 cached keys directly
    g_ = torch.Generator().manual_seed(1)
    cache[0].k = torch.randn(1, 2, window - 1, 128, generator=g_)
    cache[0].v = torch.randn(1, 2, window - 1, 128, generator=g_)
    cache[0].counters = SwaCounters(window, size=window - 1, cumulative_length=4 * window)
    x = torch.randn(1, chunk, oc.hidden_size, generator=g_) * 0.02
    pid = torch.arange(4 * window, 4 * window + chunk)[None, None, :].expand(3, 1, chunk).contiguous()
    omodel.text_stack(params, x, pid, oc, cache)                      # warm-up (also flips the GDN `start` flags)
    n, t0 = 0, time.perf_counter()
    while True:
        pid = pid + chunk
        omodel.text_stack(params, x, pid, oc, cache)
        n += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n >= 40:
            break
    per_period = el / n
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=chunk / (per_period * 9.0), unit="tok/s", cores=torch.get_num_threads(), cpu_model=cpu_model,
                host_logical_cpus=ncores, kind="port",
                sample=f"oracle/model.py on host CPU: one 4-layer period (1 SWA + 3 GDN decoder layers, fp32, real "
                       f"InfiniteVL-3B shapes) x {n} steps of {chunk} tokens with a full {window}-key window, "
                       f"{per_period:.3f} s per period-step, extrapolated x9 to 36 layers")


def _self_spawn(n: int) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: re-execute under torch.distributed.run,
    one rank per GPU of this node (rendezvous on 127.0.0.1, a free port)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_spawn(args.gpus)
    from infinitevl_amd import dist as ivd
    # IVL_DIST_BACKEND=gloo lets the N > 1 control flow be exercised with several ranks on ONE GPU (RCCL refuses two
    # ranks per device); production is always "nccl" (= RCCL over xGMI)
    rank, world, local_rank = ivd.init_distributed(os.environ.get("IVL_DIST_BACKEND", "nccl"))
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    import infinitevl_amd
    infinitevl_amd.load_library()
    from infinitevl_amd.harness import GraphedDecode, GraphedStep, InfiniteVLTextConfig, InfiniteVLTextStack

    cfg = InfiniteVLTextConfig(sliding_window=args.window, num_hidden_layers=args.layers)
    with torch.device(device):
        torch.set_default_dtype(torch.bfloat16)
        model = InfiniteVLTextStack(cfg)
        torch.set_default_dtype(torch.float32)
    model = model.to(torch.bfloat16).eval()
    model.init_weights_(seed=0)
    model.fuse_()                                  # fused projections + prologue/epilogue kernels
    B_local = 1                                    # one sequence per GPU (weak scaling over the batch)
    cache = model.allocate_inference_cache(B_local)
    T = args.chunk
    gen = torch.Generator(device=device).manual_seed(100 + rank)
    frames = [(torch.randn(B_local, T, cfg.hidden_size, device=device, generator=gen) * 0.02).to(torch.bfloat16)
              for _ in range(4)]

    step = GraphedStep(model, cache, B_local, T, logits_to_keep=1)
    step.capture()
    # untimed setup: stream up to (context - steps*T) tokens, at least the window fill, so that every timed step is a
    # steady-state full-window step and the timed region ENDS at `context` tokens (128K) whatever --steps is
    fill_min = (args.window - 1 + T - 1) // T + 1
    fill = max(fill_min, (args.context - args.steps * T) // T - args.warmup)
    for i in range(fill):
        step.step(frames[i % 4])
    for i in range(args.warmup):
        step.step(frames[i % 4])
    ivd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step.step(frames[i % 4])
    torch.cuda.synchronize()
    ivd.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = ivd.max_over_ranks(elapsed, device)
    tokens_global = args.steps * T * B_local * world
    ctx_tokens = cache.get_seq_length()

    # the path's only collective: gather the last-position logits of every sequence
    _, logits = step.hidden, step.logits
    all_logits = ivd.gather_last_logits(logits[:, -1].float().contiguous(), [B_local] * world)
    finite = bool(torch.isfinite(all_logits).all())

    # ---- decode leg (separately timed) ------------------------------------------------------------
    dec = GraphedDecode(model, cache, B_local)
    dec.token.copy_(logits[:, -1].argmax(-1, keepdim=True))
    dec.capture()
    for _ in range(4):
        dec.step()
    ivd.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.decode_steps):
        dec.step()
    torch.cuda.synchronize()
    ivd.barrier()
    dec_elapsed = ivd.max_over_ranks(time.perf_counter() - t1, device)
    mem_gb = torch.cuda.max_memory_allocated(device) / 2 ** 30

    # ---- fp8 leg (rank 0, reported beside the headline, never mixed into `value`): BASELINE.json configs[4] -- the same
    #      steady-state streaming step and decode step with e4m3 operands in the GDN chunk scan / SWA decode step
    fp8 = None
    if rank == 0 and not args.no_fp8:
        model.set_mma_dtype("fp8_e4m3")
        cache8 = cache.clone()
        step8 = GraphedStep(model, cache8, B_local, T, logits_to_keep=1)
        step8.capture()
        for i in range(4):
            step8.step(frames[i % 4])
        torch.cuda.synchronize()
        tA = time.perf_counter()
        n8 = 32
        for i in range(n8):
            step8.step(frames[i % 4])
        torch.cuda.synchronize()
        t8 = (time.perf_counter() - tA) / n8
        fin8 = bool(torch.isfinite(step8.logits.float()).all())
        dec8 = GraphedDecode(model, cache8, B_local)
        dec8.token.copy_(step8.logits[:, -1].argmax(-1, keepdim=True))
        dec8.capture()
        for _ in range(4):
            dec8.step()
        torch.cuda.synchronize()
        tA = time.perf_counter()
        for _ in range(n8):
            dec8.step()
        torch.cuda.synchronize()
        td8 = (time.perf_counter() - tA) / n8
        fp8 = {"workload": "the same steady-state 256-token streaming step and decode step with e4m3 operands in the GDN chunk "
                           "scan and the SWA decode products (fp32 accumulation and state)", "steps_timed": n8,
               "ms_per_step": t8 * 1e3, "prefill_tok_s": T / t8, "decode_ms_per_token": td8 * 1e3, "decode_tok_s": 1.0 / td8,
               "frames_per_s_at_256_tokens": 1.0 / t8, "frame_budget_ms_24fps": 1000.0 / 24, "logits_finite": fin8}
        model.set_mma_dtype(None)
        del step8, dec8, cache8

    # ---- configs[1] leg (rank 0, N=1; reported beside the headline, never mixed into `value`): one 4096-token
    #      prefill call on a fresh cache (chunk path over 64 chunks; SWA purely causal) + 128 graphed decode steps
    cfg1 = None
    if rank == 0 and world == 1 and not args.no_cfg1:
        del step, dec
        Tp, reps = 4096, 3
        ids = torch.randint(0, cfg.vocab_size, (1, Tp), device=device,
                            generator=torch.Generator(device=device).manual_seed(1))
        pid = torch.arange(Tp, device=device)[None, None, :].expand(3, 1, Tp).contiguous()
        cache1 = model.allocate_inference_cache(1)
        times = []
        with torch.no_grad():
            for r in range(reps + 1):
                cache1.reset()
                torch.cuda.synchronize()
                tA = time.perf_counter()
                _, lg = model(input_ids=ids, position_ids=pid, past_key_values=cache1, logits_to_keep=1)
                torch.cuda.synchronize()
                if r > 0:
                    times.append(time.perf_counter() - tA)
        dec1 = GraphedDecode(model, cache1, 1)
        dec1.token.copy_(lg[:, -1].argmax(-1, keepdim=True))
        dec1.capture()
        for _ in range(4):
            dec1.step()
        torch.cuda.synchronize()
        tA = time.perf_counter()
        for _ in range(128):
            dec1.step()
        torch.cuda.synchronize()
        tdec = time.perf_counter() - tA
        pf = min(times)
        cfg1 = {"workload": "configs[1]: 4096-token prefill in one call (eager launches) + 128 graphed decode steps, B=1",
                "prefill_ms": pf * 1e3, "prefill_tok_s": Tp / pf, "decode_tok_s": 128 / tdec,
                "decode_ms_per_token": tdec / 128 * 1e3, "logits_finite": bool(torch.isfinite(lg.float()).all())}
        del dec1, cache1

    kernels, cpu = None, None
    if rank == 0 and not args.no_kernel_timing:
        kernels = kernel_timings(device, T, args.window)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(T, args.window)
    ivd.barrier()

    if rank == 0:
        out = {
            "metric": "prefill tok/s + decode tok/s, InfiniteVL-3B @128K seq",
            "value": tokens_global / elapsed, "unit": "tok/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "decode_tok_s": args.decode_steps * B_local * world / dec_elapsed,
            "decode_ms_per_token": dec_elapsed / args.decode_steps * 1e3,
            "config": {
                "workload": f"InfiniteVL-3B ({args.layers} layers: 9 SWA + 27 Gated DeltaNet, random-init bf16) streaming "
                            f"prefill, {T}-token hipGraph chunk steps, SWA window {args.window} (full), then greedy decode "
                            f"at the reached context",
                "global_batch": B_local * world, "seq_len": ctx_tokens, "tokens_timed": tokens_global,
                "parallelism": f"dp{world} (batch-sharded replicas, one logits all-gather)",
                "decode_steps": args.decode_steps, "untimed_steps_before_timed_region": fill + args.warmup,
            },
            "logits_finite": finite, "peak_mem_gib": round(mem_gb, 2),
        }
        if kernels is not None:
            def step_ms(r):
                return r["ms"] * r["launches_per_step"]
            prefill_kernels = {k: v for k, v in kernels.items() if v["launches_per_step"] > 0 and "decode" not in k}
            dom = max(prefill_kernels, key=lambda k: step_ms(prefill_kernels[k]))
            r = kernels[dom]
            tr = pmc_traffic(dom, T, args.window)
            out["roofline"] = {"kernel": dom, "bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"],
                               "unit": r["unit"], "frac": r["frac"], "traffic": tr["hbm_bytes"] if tr else None,
                               "traffic_source": tr["source"] if tr else None,
                               "algorithmic_per_launch": r.get("alg_bytes", r.get("alg_flops")),
                               "avg_launch_ms": r["ms"], "launches_per_step": r["launches_per_step"]}
            out["kernels"] = {k: {kk: (round(vv, 6) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                              for k, v in kernels.items()}
            out["hot_path_ms_per_step"] = sum(step_ms(v) for v in prefill_kernels.values())
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if fp8 is not None:
            out["fp8_e4m3"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in fp8.items()}
        if cfg1 is not None:
            out["cfg1_4k_prefill_decode"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cfg1.items()}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
