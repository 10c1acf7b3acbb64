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
    ap.add_argument("--no-cfg3", action="store_true", help="skip the bulk-prefill leg (configs[3]: 4096-token calls at >= 128K context)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-fp8", action="store_true", help="skip the fp8 (e4m3) leg (BASELINE.json configs[4])")
    ap.add_argument("--cfg3-tokens", type=int, default=0,
                    help="configs[3] leg: 0 (default) = 2 + 8 calls of 4096 tokens continuing from the reached context; N > 0 = a FRESH "
                         "sequence of N tokens per rank in 4096-token calls from position 0 (524288 = the full configs[3] sequence, 128 calls)")
    ap.add_argument("--sp-tokens", type=int, default=4096, help="N > 1: tokens per rank of the sequence-parallel prefill leg")
    ap.add_argument("--no-sp", action="store_true", help="N > 1: skip the sequence-parallel prefill leg (SURVEY.md 8f-4)")
    return ap.parse_args()


def event_time_ms(fn, iters, stream):
    """Average GPU duration of `fn(i)`: `iters` calls (i = 0 .. iters-1) are captured into one hipGraph (so the measurement
    is not bound by the Python/ctypes launch overhead of ~10-30 us per call) and the replay is timed with HIP events
    recorded on the stream the kernels run on.  `fn` picks its input set from i: the step-phase entries rotate over input
    sets that together exceed the L2s, so a launch does not find the previous launch's lines there (in the real step the
    inputs of a kernel were written by the GEMM in front of it, not read by its own previous launch)."""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    from infinitevl_amd import ops as _ops
    _ops.prepare_gdn_capture()             # scope-less capture of (possibly large) GDN chunk calls: their records workspace must exist
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(iters):
            fn(i)
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


# rocprofv3 kernel-trace rows (kernel-name prefix, grid threads) that make up ONE call of an entry of `kernels`: used to
# attach the committed in-step durations (profiles/rNN_bench_kernel_by_grid.csv) and PMC traffic (rNN_pmc_traffic.json)
PROFILE_ROWS = {
    "gdn_chunk(prepare+scan)": [("ivl::gdn_chunk_prepare_kernel<false, false>", 32768), ("ivl::gdn_chunk_scan_kernel<2, false, false>", 98304)],
    # one launch at the step shape: 2 x 64 pre-pass (k side, q side) + 128 scan workgroups of 768 threads
    "gdn_chunk_fused(convs+gates+prepare+scan)": [("ivl::gdn_chunk_single_kernel<false, 1>", 196608)],
    "swa_prefill(rope pre-pass + attention + combine with append)": [("ivl::swa_rope_prepass_kernel", 36864),
                                                                     ("ivl::swa_prefill_kernel", 196608),
                                                                     ("ivl::swa_combine_kernel<8, true>", 270336)],
    "gdn_prologue(3 convs + gates)": [("ivl::gdn_prologue_kernel", 69632)],
    "add_rmsnorm(decoder layer)": [("ivl::add_rmsnorm_kernel", 65536)],
    "rmsnorm_swish_gate": [("ivl::rmsnorm_gate_strided_kernel", 131072)],
    "silu_mul(SwiGLU gate)": [("ivl::silu_mul_kernel", 352256)],
    "rope_tables": [("ivl::rope_tables_kernel", 49152)],
    "counter_add": [("ivl::counter_add_kernel", 64)],
    "gdn_recurrent(decode)": [("ivl::gdn_recurrent_kernel", 32768)],
    "gdn_decode_step(decode: convs+gates+rule+norm, 1 launch)": [("ivl::gdn_decode_step_kernel", 8192)],
    "swa_decode": [("ivl::swa_fwd_kernel<true, 1>", 32768), ("ivl::swa_combine_wide_kernel", 4352)],
    "add_rmsnorm(decode token)": [("ivl::add_rmsnorm_kernel", 256)],
}


def _newest_profile(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def in_step_us(kernel_name, chunk, window):
    """Duration of one call of a `kernels` entry INSIDE the real step / decode token: the sum over its launches of the
    (kernel, grid) averages of the newest committed rocprofv3 trace of this same command
    (profiles/rNN_bench_kernel_by_grid.csv, tools/collect_profiles.sh).  None when the shapes differ from the profiled
    ones or a row is missing (kernel renamed / regridded since the profile was taken)."""
    import csv
    want = PROFILE_ROWS.get(kernel_name)
    path = _newest_profile("r*_bench_kernel_by_grid.csv")
    if not want or path is None or chunk != 256 or window != 4096:
        return None
    rows = {(r["kernel"], int(r["grid_threads"])): float(r["avg_ns"]) for r in csv.DictReader(open(path))}
    total = 0.0
    for base, grid in want:
        hit = [v for (n, g), v in rows.items() if g == grid and n.startswith(base)]
        if len(hit) != 1:
            return None
        total += hit[0]
    return {"us": total / 1e3, "source": os.path.basename(path)}


def profile_replays(replay, n=4):
    """Per-kernel GPU durations INSIDE a captured step, measured live: `n` replays of the hipGraph under torch.profiler (the
    rocprofiler activity records of the replayed kernels).  Returns {kernel name without arguments: (avg_us, launches per
    replay)}.  This is the in-process twin of tools/collect_profiles.sh's rocprofv3 --kernel-trace of the same command."""
    import collections
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            replay()
        torch.cuda.synchronize()
    agg = collections.defaultdict(list)
    for ev in prof.events():
        if "cuda" in str(ev.device_type).lower():
            dur = ev.device_time if hasattr(ev, "device_time") else ev.cuda_time
            agg[ev.name.split("(")[0].replace("void ", "").strip()].append(float(dur))
    return {k: (sum(v) / len(v), len(v) / n) for k, v in agg.items()}


def live_in_step_us(kernel_name, live):
    """One call of a `kernels` entry inside the profiled graph: the sum of the live averages of its launches (PROFILE_ROWS)."""
    want = PROFILE_ROWS.get(kernel_name)
    if not want or not live:
        return None
    total = 0.0
    for base, _grid in want:
        hit = [v[0] for n, v in live.items() if n.startswith(base)]
        if len(hit) != 1:
            return None
        total += hit[0]
    return total


def kernel_timings(device, chunk, window, only=None, live_prefill=None, live_decode=None):
    """Per-call GPU time of every hot-path kernel at the bench shapes (B=1, T=chunk, InfiniteVL-3B head shapes, full
    window), plus throughput-regime shapes (B=8, T=4096) for the two heavy kernels.  Each entry: `phase` ("prefill":
    part of the streaming step, "decode": part of a decode token, "other": not in either), ms per call, calls per step /
    token, the roofline that bounds it, algorithmic bytes / flops per call and the achieved rate.
    Step- and token-phase entries are timed over rotating input sets (see event_time_ms)."""
    from infinitevl_amd import ops
    st = torch.cuda.current_stream(device)
    B, T, H, K, V, Hq, Hkv, d = 1, chunk, 16, 128, 256, 16, 2, 128
    g_ = torch.Generator(device=device).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=device, generator=g_).to(torch.bfloat16)  # noqa: E731
    C = window - 1
    res = {}
    NS = 6                                         # input sets a step-phase entry rotates over

    def add(name, fn, iters, phase, per, bound, work):
        if only == "!large":                       # the PMC passes: bench-shape launches only (keyed by kernel + grid)
            if "@T=" in name or "@B=" in name:
                return
        elif only is not None and only not in name:
            return
        ms = event_time_ms(fn, iters, st)
        if bound == "hbm":
            res[name] = dict(phase=phase, ms=ms, bound="hbm", alg_bytes=work,
                             achieved=work / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        else:
            res[name] = dict(phase=phase, ms=ms, bound="mfma", alg_flops=work,
                             achieved=work / (ms * 1e-3) / 1e12, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s")
        res[name]["launches_per_step"] = per if phase == "prefill" else 0
        res[name]["launches_per_decode_token"] = per if phase == "decode" else 0
        res[name]["frac"] = res[name]["achieved"] / res[name]["peak"]
        # the same call inside the real step / decode token: measured live in this run (torch.profiler over replays of the
        # captured graph) and, as a cross-check, from the committed rocprofv3 trace of this command
        live = live_in_step_us(name, live_prefill if phase == "prefill" else live_decode if phase == "decode" else None)
        ins = in_step_us(name, chunk, window)
        if ins is not None:
            res[name]["in_step_us_rocprofv3"] = ins["us"]
            res[name]["in_step_rocprofv3_source"] = ins["source"]
        if live is not None or ins is not None:
            us = live if live is not None else ins["us"]
            res[name]["in_step_us"] = us
            res[name]["in_step_source"] = "live: torch.profiler over replays of the captured graph" if live is not None else ins["source"]
            res[name]["frac_in_step"] = work / (us * 1e-6) / (1e9 if bound == "hbm" else 1e12) / res[name]["peak"]

    def gdn_inputs(Tn, Bn=B):
        q, k, v = rn(Bn, Tn, H, K), rn(Bn, Tn, H, K), rn(Bn, Tn, H, V)
        beta = torch.rand(Bn, Tn, H, device=device, generator=g_).to(torch.bfloat16)
        g = torch.nn.functional.logsigmoid(torch.randn(Bn, Tn, H, device=device, generator=g_))
        return q, k, v, g, beta

    states = [torch.randn(B, H, K, V, device=device, generator=g_).to(torch.bfloat16) for _ in range(NS)]
    state = states[0]
    # GDN bytes: 24,672 B/token/layer + state read+write per call in the dtype actually passed (bf16 cache: SURVEY.md Q5)
    sbytes = 2 * H * K * V * state.element_size()
    gin = [gdn_inputs(T) for _ in range(NS)]
    add("gdn_chunk(prepare+scan)", lambda i: ops.chunk_gated_delta_rule(
        *gin[i % NS], initial_state=states[i % NS], use_qk_l2norm_in_kernel=True, final_state_out=states[i % NS]),
        24, "other", 0, "hbm", 24672.0 * T + sbytes)
    add("gdn_chunk_fp8(prepare+scan)", lambda i: ops.chunk_gated_delta_rule(
        *gin[i % NS], initial_state=states[i % NS], use_qk_l2norm_in_kernel=True, final_state_out=states[i % NS],
        mma_dtype="fp8_e4m3"), 24, "other", 0, "hbm", 24672.0 * T + sbytes)
    del gin
    # the step's GDN layers run the chunk kernel WITH its front end (3 convs + SiLU + gate math inside the kernels): the
    # projection is read once (16,416 B/token incl. the a / b gate inputs) and q / k / v / g / beta never reach HBM;
    # algorithmic bytes per token: 16,416 (projection slice) + 8,192 (o) = 24,608 B + the state
    Dq, Dk, Dv = H * K, H * K, H * V
    cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
    ld = cols[4] + H
    projs = [rn(B, T, ld) for _ in range(NS)]
    cw = [rn(D_, 1, 4) for D_ in (Dq, Dk, Dv)]
    css = [[rn(B, D_, 4) for D_ in (Dq, Dk, Dv)] for _ in range(NS)]
    A32, dt32 = torch.randn(H, device=device, generator=g_), torch.randn(H, device=device, generator=g_)
    add("gdn_chunk_fused(convs+gates+prepare+scan)", lambda i: ops.gdn_chunk_fused(
        projs[i % NS], cols, cw, css[i % NS], css[i % NS], A32, dt32, H, K, V, initial_state=states[i % NS],
        final_state_out=states[i % NS]), 24, "prefill", 27, "hbm", 24608.0 * T + sbytes)
    add("gdn_prologue(3 convs + gates)", lambda i: ops.gdn_prologue(projs[i % NS], cols, cw, css[i % NS], css[i % NS], A32, dt32,
                                                                    H, Dq, Dk, Dv),
        48, "other", 0, "hbm", 2.0 * T * 8192 * 2 + T * H * (2 * 2 + 4 + 2))
    xos, wn = [rn(B, T, H, V) for _ in range(NS)], rn(V)
    add("rmsnorm_swish_gate", lambda i: ops.rmsnorm_swish_gate_strided(xos[i % NS], projs[i % NS][..., Dq + Dk + Dv:], ld, wn, 1e-5),
        48, "prefill", 27, "hbm", 3.0 * T * H * V * 2)
    # SWA layer of the step, called the way cache.attend calls it (cache.py: StaticSlidingWindowLayerPrealloc.attend): q / k /
    # v are strided views of the fused qkv projection, M-RoPE fused (rope=), ring append in the combine launch (append=True):
    # rope pre-pass + attention + combine are all inside.  T queries over a full ring (W-1 cached keys) + T new keys;
    # 8192*min(p+1,W) FLOP/token/layer
    NSW = 4
    rings = [(rn(B, Hkv, C, d), rn(B, Hkv, C, d)) for _ in range(NSW)]
    pos_dev = torch.full((1,), 10 * window, dtype=torch.int64, device=device)
    nq, nkv = Hq * d, Hkv * d

    def qkv_views(Tn, Bn=B):
        qkv = rn(Bn, Tn, nq + 2 * nkv)
        return (qkv[..., :nq].unflatten(-1, (Hq, d)), qkv[..., nq:nq + nkv].unflatten(-1, (Hkv, d)),
                qkv[..., nq + nkv:].unflatten(-1, (Hkv, d)))
    qkvs = [qkv_views(T) for _ in range(NSW)]
    pid = torch.arange(10 * window, 10 * window + T, device=device)[None, None, :].expand(3, B, T).contiguous()
    inv_freq = 1.0 / (1e6 ** (torch.arange(0, d, 2, device=device, dtype=torch.float32) / d))
    cos, sin = ops.rope_tables(pid, inv_freq, 1.0)
    rope = (cos, sin, (16, 24, 24))
    add("swa_prefill(rope pre-pass + attention + combine with append)", lambda i: ops.swa_forward(
        *qkvs[i % NSW], window=window, scaling=d ** -0.5, k_cache=rings[i % NSW][0], v_cache=rings[i % NSW][1],
        pos_dev=pos_dev, rope=rope, append=True), 24, "prefill", 9, "mfma", 4.0 * Hq * d * window * T)
    add("swa_attention_only(rotated inputs, no append)", lambda i: ops.swa_forward(
        *qkvs[i % NSW], window=window, scaling=d ** -0.5, k_cache=rings[i % NSW][0], v_cache=rings[i % NSW][1],
        pos_dev=pos_dev), 24, "other", 0, "mfma", 4.0 * Hq * d * window * T)
    add("swa_cache_append(stand-alone launch; the step appends in the combine launch)", lambda i: ops.swa_cache_append(
        qkvs[i % NSW][1], qkvs[i % NSW][2], rings[i % NSW][0], rings[i % NSW][1], pos_dev=pos_dev), 48, "other", 0, "hbm",
        4.0 * T * Hkv * d * 2)
    add("rope_tables", lambda i: ops.rope_tables(pid, inv_freq, 1.0), 48, "prefill", 1, "hbm", 3.0 * T * (8 + 2 * 2 * d))
    add("counter_add", lambda i: ops.counter_add(pos_dev, 0), 48, "prefill", 1, "hbm", 16.0)
    # decoder-layer glue: residual add + RMSNorm (2 per layer + the final norm), SwiGLU gate
    xhs, rhs, wh = [rn(B, T, 2048) for _ in range(NS)], [rn(B, T, 2048) for _ in range(NS)], rn(2048)
    add("add_rmsnorm(decoder layer)", lambda i: ops.add_rmsnorm(xhs[i % NS], rhs[i % NS], wh, 1e-6), 48, "prefill", 73, "hbm",
        4.0 * T * 2048 * 2)
    I_ = 11008
    gus = [rn(B, T, 2 * I_) for _ in range(NS)]
    add("silu_mul(SwiGLU gate)", lambda i: ops.silu_mul(gus[i % NS]), 48, "prefill", 36, "hbm", 3.0 * T * I_ * 2)
    del gus, xhs, rhs, xos
    # decode-shape kernels (per decode token)
    q1, k1, v1, g1, b1 = gdn_inputs(1)
    add("gdn_recurrent(decode)", lambda i: ops.fused_recurrent_gated_delta_rule(
        q1, k1, v1, g1, b1, initial_state=states[i % NS], use_qk_l2norm_in_kernel=True, final_state_out=states[i % NS]),
        96, "other", 0, "hbm", 24672.0 + sbytes)
    proj1 = rn(B, 1, ld)
    cols6 = (cols[0], cols[1], cols[2], Dq + Dk + Dv, cols[3], cols[4])
    add("gdn_decode_step(decode: convs+gates+rule+norm, 1 launch)", lambda i: ops.gdn_decode_step(
        proj1, cols6, cw, css[i % NS], A32, dt32, wn, 1e-5, states[i % NS], H, K, V, K ** -0.5), 96, "decode", 27, "hbm",
        2.0 * ld + sbytes + 2.0 * H * V)
    # the measured alternative (round 5, off by default: ops._SPLIT_DECODE): the step on 64 workgroups + the gated norm and the q / k
    # conv-state shift in the o_proj launch -- 1.953 vs 1.925 ms per token against the one-launch step + plain o_proj
    wo_ = rn(2048, H * V)
    add("gdn_decode_split(64 workgroups, un-normalised output)", lambda i: ops.gdn_decode_split(
        proj1, (cols6[0], cols6[1], cols6[2], cols6[4], cols6[5]), cw, css[i % NS], A32, dt32, states[i % NS], H, K, V, K ** -0.5),
        96, "other", 0, "hbm", 2.0 * ld + sbytes + 2.0 * H * V)
    oraw_ = rn(B, 1, H * V)
    add("decode linear gated norm + gdn o_proj [2048x4096] (+ q/k conv-state shift)", lambda i: ops.gdn_out_linear(
        oraw_, proj1, cols6[3], cols6[0], cols6[1], wn, 1e-5, css[i % NS][0], css[i % NS][1], wo_, None, H), 96, "other", 0, "hbm",
        2.0 * 2048 * H * V + 2.0 * (2048 + 2 * H * V))
    del wo_
    qd, kd1, vd1 = qkv_views(1)
    add("swa_decode", lambda i: ops.swa_forward(qd, kd1, vd1, window=window, scaling=d ** -0.5, k_cache=rings[i % NSW][0],
                                                v_cache=rings[i % NSW][1], pos_dev=pos_dev, append=True), 96, "decode", 9, "hbm",
        1024.0 * window)
    x1, r1 = rn(B, 1, 2048), rn(B, 1, 2048)
    # (round 3: the 72 per-layer norms of a decode token run in the prologue of the projection that follows them -- only the
    # final norm is still a launch of its own)
    add("add_rmsnorm(decode token)", lambda i: ops.add_rmsnorm(x1, r1, wh, 1e-6), 96, "decode", 1, "hbm", 4.0 * 2048 * 2)
    # decode-step projections (M = 1 weight streams; outside SURVEY.md section 8's rows, listed for the decode leg); "norm +":
    # (residual add +) RMSNorm in the kernel's prologue (ivl_norm_linear_small_m_fwd)
    for nm, N_, K_, per, norm in (("norm + gdn in-proj", 12320, 2048, 27, True), ("norm + mlp gate|up (SwiGLU)", 11008, 2048, 36, True),
                                  ("mlp down", 2048, 11008, 36, False), ("gdn o_proj", 2048, 4096, 27, False),
                                  ("lm_head", 151936, 2048, 1, False)):
        glu = "SwiGLU" in nm
        w_, x_ = rn(2 * N_ if glu else N_, K_), rn(1, 1, K_)
        if norm:
            fn = (lambda i, w_=w_, x_=x_: ops.linear_swiglu(ops.PreNorm(x_, r1, wh, 1e-6), w_)) if glu else \
                 (lambda i, w_=w_, x_=x_: ops.linear(ops.PreNorm(x_, r1, wh, 1e-6), w_))
        else:
            fn = lambda i, w_=w_, x_=x_: ops.linear(x_, w_)      # noqa: E731
        add(f"decode linear {nm} [{(2 * N_ if glu else N_)}x{K_}]", fn, 48, "decode", per, "hbm",
            2.0 * (2 * N_ if glu else N_) * K_ + 2.0 * (N_ + K_))
        del w_, x_
    del projs, css
    # batched streams (8 sequences per GPU in one call): the same kernels with 8x the work per launch
    Bb = 8
    ginb = gdn_inputs(T, Bb)
    stateb = torch.randn(Bb, H, K, V, device=device, generator=g_).to(torch.bfloat16)
    add("gdn_chunk@B=8", lambda i: ops.chunk_gated_delta_rule(
        *ginb, initial_state=stateb, use_qk_l2norm_in_kernel=True, final_state_out=stateb),
        10, "other", 0, "hbm", Bb * (24672.0 * T + sbytes))
    add("gdn_chunk_fp8@B=8", lambda i: ops.chunk_gated_delta_rule(
        *ginb, initial_state=stateb, use_qk_l2norm_in_kernel=True, final_state_out=stateb,
        mma_dtype="fp8_e4m3"), 10, "other", 0, "hbm", Bb * (24672.0 * T + sbytes))
    projb = rn(Bb, T, ld)
    csb = [rn(Bb, D_, 4) for D_ in (Dq, Dk, Dv)]
    add("gdn_chunk_fused@B=8", lambda i: ops.gdn_chunk_fused(projb, cols, cw, csb, csb, A32, dt32, H, K, V, initial_state=stateb,
                                                             final_state_out=stateb), 10, "other", 0, "hbm",
        Bb * (24608.0 * T + sbytes))
    ringb = (rn(Bb, Hkv, C, d), rn(Bb, Hkv, C, d))
    qkvb = qkv_views(T, Bb)
    pidb = pid.expand(3, Bb, T).contiguous()
    cosb, sinb = ops.rope_tables(pidb, inv_freq, 1.0)
    add("swa_prefill@B=8", lambda i: ops.swa_forward(*qkvb, window=window, scaling=d ** -0.5, k_cache=ringb[0], v_cache=ringb[1],
                                                     pos_dev=pos_dev, rope=(cosb, sinb, (16, 24, 24)), append=True),
        10, "other", 0, "mfma", Bb * 4.0 * Hq * d * window * T)
    del ginb, projb, csb, ringb, qkvb
    # throughput-regime shapes: 4096-token calls (BASELINE.json configs[1]: fresh cache, causal; configs[3]: bulk prefill over
    # a FULL ring at >= 128K context, the cfg3_512k_prefill leg); not part of the step
    TL = 4096
    ginL = gdn_inputs(TL)
    add("gdn_chunk@T=4096", lambda i: ops.chunk_gated_delta_rule(
        *ginL, initial_state=state, use_qk_l2norm_in_kernel=True, final_state_out=state),
        5, "other", 0, "hbm", 24672.0 * TL + sbytes)
    add("gdn_chunk_fp8@T=4096", lambda i: ops.chunk_gated_delta_rule(
        *ginL, initial_state=state, use_qk_l2norm_in_kernel=True, final_state_out=state, mma_dtype="fp8_e4m3"),
        5, "other", 0, "hbm", 24672.0 * TL + sbytes)
    projL = rn(B, TL, ld)
    csL = [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)]
    add("gdn_chunk_fused@T=4096", lambda i: ops.gdn_chunk_fused(projL, cols, cw, csL, csL, A32, dt32, H, K, V, initial_state=state,
                                                                final_state_out=state), 5, "other", 0, "hbm",
        24608.0 * TL + sbytes)
    qkvL = qkv_views(TL)
    add("swa_prefill@T=4096(causal)", lambda i: ops.swa_forward(*qkvL, window=8192, scaling=d ** -0.5),
        5, "other", 0, "mfma", 4.0 * Hq * d * (TL * (TL + 1) / 2))
    pidL = torch.arange(40 * window, 40 * window + TL, device=device)[None, None, :].expand(3, B, TL).contiguous()
    cosL, sinL = ops.rope_tables(pidL, inv_freq, 1.0)
    add("swa_prefill@T=4096(full ring, rope + append)", lambda i: ops.swa_forward(
        *qkvL, window=window, scaling=d ** -0.5, k_cache=rings[0][0], v_cache=rings[0][1], pos_dev=pos_dev,
        rope=(cosL, sinL, (16, 24, 24)), append=True), 5, "other", 0, "mfma", 4.0 * Hq * d * window * TL)
    # the same call on the 256-row kernel (swa_ring256.hip: linearize pre-pass with the append + attention on the linear keys) --
    # what cache.attend selects outside a capture once the ring is full; the entry above stays on the 128-row kernel
    add("swa_ring256@T=4096(full ring, rope + append)", lambda i: ops.swa_forward(
        *qkvL, window=window, scaling=d ** -0.5, k_cache=rings[0][0], v_cache=rings[0][1], pos_dev=pos_dev,
        rope=(cosL, sinL, (16, 24, 24)), append=True, pos_min=10 * window, pos_min_holds_in_graph=True),
        5, "other", 0, "mfma", 4.0 * Hq * d * window * TL)
    del ginL, projL, qkvL
    # vision tower (not part of the text-stack step): 8 frames of 32 x 32 patches, 16 heads x 80,
    # rotary embedding folded in; a window layer (64-patch segments) and a full-attention layer (one segment per frame).
    # flops = 4 * d * H * sum(len^2) (non-causal)
    Hv, dv, frames, per = 16, 80, 8, 1024
    qkv = rn(frames * per, 3, Hv, dv)
    vcos, vsin = (torch.randn(frames * per, dv, device=device, generator=g_) for _ in range(2))
    # a window layer does 4 * 80 * 64 flops per byte-pair of its q / k / v / o rows: ~41 flop per byte of HBM traffic, far
    # left of the ridge (2,500 TFLOP/s / 8 TB/s = 312 flop / byte) -- HBM-bound: q, k, v read + o written (bf16) + the fp32
    # rotary tables; the full-attention layer (1024-patch segments: 16x the flops on the same bytes) is MFMA-bound
    vis_bytes = frames * per * (4 * Hv * dv * 2 + 2 * dv * 4)
    for tag, seg in (("window layer, 64-patch segments", 64), ("full layer, 1024-patch segments", 1024)):
        cu = torch.arange(0, frames * per + 1, seg, dtype=torch.int32, device=device)
        vis_flops = 4.0 * dv * Hv * (frames * per // seg) * seg * seg
        hbm_bound = vis_flops / vis_bytes < MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        add(f"vision_attn({tag})@8 frames", lambda i, cu=cu, seg=seg: ops.vision_window_attention(
            qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, seg, rope=(vcos, vsin)), 20, "other", 0,
            "hbm" if hbm_bound else "mfma", vis_bytes if hbm_bound else vis_flops)
    return res


def pmc_traffic(kernel_name, chunk, window):
    """HBM-side bytes per call of a hot-path kernel from the committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_traffic.json, produced by tools/collect_profiles.sh on this same command's kernels at the
    default bench shapes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot
    run the PMC passes itself, so the newest committed measurement is attached; null when the shapes differ."""
    if chunk != 256 or window != 4096:
        return None
    path = _newest_profile("r*_pmc_traffic.json")
    want = PROFILE_ROWS.get(kernel_name)
    if path is None or not want:
        return None
    k = json.load(open(path))["kernels"]
    parts = []
    for base, grid in want:          # kernel names carry template arguments: match on the prefix and the grid size
        hit = [n for n in k if n.startswith(base) and n.endswith(f"@grid{grid}")]
        if len(hit) != 1:
            return None
        parts.append(hit[0])
    return {"hbm_bytes": sum(k[p_]["hbm_bytes"] for p_ in parts), "source": os.path.basename(path),
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

    def all_ok(ok: bool) -> bool:
        """Exchange an ok flag (all_reduce MIN) after every stage that can fail on ONE rank (an allocation, a kernel error), BEFORE
        the next collective: a rank that raised then does not leave the others blocked in a barrier / all_reduce -- every rank
        skips the rest of that leg and the error is recorded."""
        if world == 1:
            return ok
        t_ = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        torch.distributed.all_reduce(t_, op=torch.distributed.ReduceOp.MIN)
        return bool(int(t_.item()))

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
    last = logits[:, -1].float().contiguous()
    all_logits = ivd.gather_last_logits(last, [B_local] * world)              # first call: communicator set-up included
    torch.cuda.synchronize()
    tg = time.perf_counter()
    all_logits = ivd.gather_last_logits(last, [B_local] * world)
    torch.cuda.synchronize()
    allgather_ms = ivd.max_over_ranks((time.perf_counter() - tg) * 1e3, device)
    finite = bool(torch.isfinite(all_logits).all())
    # what the collective actually ran on (N > 1): backend, one record per rank, the gathered shape.  Ranks must sit on
    # distinct GPUs under RCCL (the gloo debug mode IVL_DIST_BACKEND=gloo may stack ranks on one device).
    dist_info = None
    if world > 1:
        dist_info = ivd.describe_ranks(device)
        dist_info.update(gathered_logits_shape=list(all_logits.shape), allgather_ms=allgather_ms,
                         logits_rows_differ_across_ranks=bool(world < 2 or not torch.equal(all_logits[0], all_logits[1])))
        if dist_info["backend"] == "nccl":
            assert dist_info["distinct_devices"] == world, dist_info
        assert tuple(all_logits.shape) == (B_local * world, cfg.vocab_size), all_logits.shape

    # per-kernel durations inside the real step, live (rank 0; outside the timed region; the cache advances like in a step)
    live_prefill = live_decode = None
    if rank == 0 and not args.no_kernel_timing:
        try:
            live_prefill = profile_replays(lambda: step.step(frames[0]))
        except Exception as e:                       # profiler unavailable: the committed rocprofv3 trace is used instead
            print(f"[bench] torch.profiler unavailable ({type(e).__name__}: {e})", file=sys.stderr)

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
    if rank == 0 and live_prefill is not None:
        try:
            live_decode = profile_replays(dec.step, n=8)
        except Exception as e:
            print(f"[bench] torch.profiler unavailable for the decode graph ({type(e).__name__}: {e})", file=sys.stderr)
    mem_gb = torch.cuda.max_memory_allocated(device) / 2 ** 30

    # ---- sequence-parallel prefill leg (N > 1; reported beside the headline, never mixed into `value`): ONE
    #      sequence of N x sp_tokens tokens cut into N consecutive segments, rank r runs the whole stack on segment r and hands
    #      every layer's carried state to rank r + 1 (point-to-point: device tensors over xGMI under RCCL, host-staged under the
    #      gloo debug backend).  Checked on the last rank against its own single-rank run of the same call sequence.
    sp = None
    if world > 1 and not args.no_sp:
        # Every stage that can fail on ONE rank (allocations, the last rank's single-rank reference run) ends in an exchange of
        # an ok flag (all_reduce MIN) BEFORE the next collective, so a rank that raised does not leave the others blocked in a
        # barrier / broadcast: all ranks then skip the rest of the leg and the error is recorded.  The headline above is
        # already measured; the leg never raises.
        Ts = args.sp_tokens
        err, xs_all, cache_sp, lg_sp, sp_ms = None, None, None, None, []
        first, last_tok = ivd.segment_bounds(world * Ts, rank, world)
        try:
            gsp = torch.Generator(device=device).manual_seed(4242)       # the same sequence on every rank
            xs_all = (torch.randn(1, world * Ts, cfg.hidden_size, device=device, generator=gsp) * 0.02).to(torch.bfloat16)
            cache_sp = model.allocate_inference_cache(1)
        except Exception as e:               # noqa: BLE001
            err = f"setup: {type(e).__name__}: {e}"
        if not all_ok(err is None):
            err = err or "setup failed on another rank"
        else:
            for rep in range(2):                                          # rep 0: communicator set-up and warm-up, rep 1: timed
                cache_sp.reset()
                ivd.barrier()
                torch.cuda.synchronize()
                tA = time.perf_counter()
                try:
                    with torch.no_grad():
                        _, lg_sp = ivd.sequence_parallel_prefill(model, xs_all[:, first:last_tok], cache_sp, first, rank, world, logits_to_keep=1)
                    torch.cuda.synchronize()
                except Exception as e:       # noqa: BLE001
                    err = f"prefill: {type(e).__name__}: {e}"
                # every rank exchanges the flag here, raised or not, BEFORE the barrier / MAX-reduce below: a rank
                # that failed in its own kernels after the hand-off (OOM, IVL_ERR_SYNC) must not leave the healthy ranks alone in
                # a collective.  (A failure INSIDE the hand-off itself is a communicator failure; only its time-out ends that.)
                if not all_ok(err is None):
                    err = err or "the prefill failed on another rank"
                    break
                ivd.barrier()
                sp_ms.append(ivd.max_over_ranks((time.perf_counter() - tA) * 1e3, device))
        check = {"equal": None, "max_abs_diff": None}
        if err is None:
            if rank == world - 1:                                         # the single-rank run of the same call sequence
                try:
                    with torch.no_grad():
                        cache_1 = model.allocate_inference_cache(1)
                        tB = time.perf_counter()
                        for r_ in range(world):
                            f_, l_ = ivd.segment_bounds(world * Ts, r_, world)
                            pos_ = torch.arange(f_, l_, device=device)[None, None, :].expand(3, 1, l_ - f_).contiguous()
                            _, lg_1 = model(inputs_embeds=xs_all[:, f_:l_], position_ids=pos_, past_key_values=cache_1, logits_to_keep=1)
                        torch.cuda.synchronize()
                        one_rank_ms = (time.perf_counter() - tB) * 1e3
                        check = {"equal": bool(torch.equal(lg_sp, lg_1)), "max_abs_diff": float((lg_sp.float() - lg_1.float()).abs().max()),
                                 "single_rank_ms": one_rank_ms, "finite": bool(torch.isfinite(lg_sp.float()).all())}
                        del cache_1
                except Exception as e:       # noqa: BLE001
                    err = f"single-rank reference run: {type(e).__name__}: {e}"
            if not all_ok(err is None):
                err = err or "the single-rank reference run failed on the last rank"
            else:
                box = [check]
                torch.distributed.broadcast_object_list(box, src=world - 1)
                check = box[0]
        if err is not None:
            sp = {"error": err, "failed": True}
        else:
            backend = torch.distributed.get_backend()
            sp = {"workload": f"sequence-parallel prefill of ONE {world * Ts}-token sequence over {world} ranks ({Ts} tokens per rank), "
                              f"per-layer carried-state hand-off rank r -> r + 1 (SURVEY.md 8f-4; the reference has no such path)",
                  "tokens": world * Ts, "ms": sp_ms[-1], "ms_first_run_with_setup": sp_ms[0], "tok_s": world * Ts / (sp_ms[-1] * 1e-3),
                  "backend": backend, "device_p2p": backend == "nccl",
                  "transport": "device tensors, batch_isend_irecv (RCCL over xGMI)" if backend == "nccl" else "host-staged (gloo debug backend)",
                  "last_token_logits_equal_single_rank_run": check["equal"], "max_abs_diff": check["max_abs_diff"],
                  # the same kernels run the same call sequence on both sides: anything but bit-equality is a failure of the hand-off
                  "failed": check["equal"] is not True,
                  "single_rank_ms_same_calls": check.get("single_rank_ms"), "speedup_vs_single_rank": (check["single_rank_ms"] / sp_ms[-1]) if check.get("single_rank_ms") else None,
                  "logits_finite": check.get("finite")}
        del cache_sp, xs_all

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

    # ---- configs[3] leg (EVERY rank: "512K-token long-context prefill, batch=8 sharded data-parallel across 8 x MI355X" = one
    #      sequence per GPU; reported beside the headline): bulk prefill of one long sequence per rank in 4096-token calls over a
    #      FULL 4096-key ring with carried GDN state (SURVEY.md 8d cfg4 "128 calls of T=4096, state carried"; reference claim
    #      README.md:51).  Eager launches.  Default: 2 + 8 calls continuing from the >= 128K-token context the legs above left in
    #      `cache`; --cfg3-tokens N: a FRESH sequence of N tokens per rank from position 0 (524288 = the whole configs[3] job).
    #      The timed region is bracketed by barriers, the slowest rank defines it (MAX over ranks), the aggregate is all ranks'
    #      tokens over that time, and the last-position logits of all sequences are gathered ONCE at the end (the path's only
    #      collective).
    cfg3 = None
    if not args.no_cfg3:
        Tb = 4096
        fresh = args.cfg3_tokens > 0
        n_warm, n_timed = (1, max(1, args.cfg3_tokens // Tb)) if fresh else (2, 8)
        err3, xb, cache3, lg3, times3, mem_marks = None, None, None, None, [], []
        try:
            xb = (torch.randn(1, Tb, cfg.hidden_size, device=device, generator=gen) * 0.02).to(torch.bfloat16)
            cache3 = model.allocate_inference_cache(1) if fresh else cache
        except Exception as e:               # noqa: BLE001
            err3 = f"setup: {type(e).__name__}: {e}"
        region3 = None
        if not all_ok(err3 is None):
            err3 = err3 or "setup failed on another rank"
        else:
            def call3():
                start3 = cache3.get_seq_length()
                pid3 = torch.arange(start3, start3 + Tb, device=device)[None, None, :].expand(3, 1, Tb).contiguous()
                return model(inputs_embeds=xb, position_ids=pid3, past_key_values=cache3, logits_to_keep=1)[1]
            try:
                with torch.no_grad():
                    for _ in range(n_warm):              # M = 4096 library GEMMs are tuned / loaded here, not in the timed calls
                        call3()
                    if fresh:
                        cache3.reset()
                    torch.cuda.synchronize()
            except Exception as e:           # noqa: BLE001
                err3 = f"warm-up: {type(e).__name__}: {e}"
            if not all_ok(err3 is None):
                err3 = err3 or "warm-up failed on another rank"
            else:
                ctx0 = cache3.get_seq_length()
                ivd.barrier()
                torch.cuda.synchronize()
                tR = time.perf_counter()
                try:
                    with torch.no_grad():
                        for r in range(n_timed):
                            tA = time.perf_counter()
                            lg3 = call3()
                            torch.cuda.synchronize()
                            times3.append(time.perf_counter() - tA)
                            if r in (1, n_timed - 1):    # after the ring is full (second call) and at the end: constant memory
                                mem_marks.append(torch.cuda.memory_allocated(device))
                except Exception as e:       # noqa: BLE001
                    err3 = f"call {len(times3)}: {type(e).__name__}: {e}"
                if not all_ok(err3 is None):
                    err3 = err3 or "a timed call failed on another rank"
                else:
                    ivd.barrier()
                    region3 = ivd.max_over_ranks(time.perf_counter() - tR, device)
        if err3 is not None:
            cfg3 = {"error": err3, "failed": True}
        else:
            mean3 = sum(times3) / len(times3)
            full = times3[1:] if fresh and len(times3) > 1 else times3     # fresh: the first call is causal over an empty ring
            per_rank_ms = [mean3 * 1e3]
            gathered3 = None
            if world > 1:
                box = [None] * world
                torch.distributed.all_gather_object(box, mean3 * 1e3)
                per_rank_ms = [float(v) for v in box]
                tg3 = time.perf_counter()
                gathered3 = ivd.gather_last_logits(lg3[:, -1].float().contiguous(), [1] * world)
                torch.cuda.synchronize()
                gather3_ms = ivd.max_over_ranks((time.perf_counter() - tg3) * 1e3, device)
                assert tuple(gathered3.shape) == (world, cfg.vocab_size), gathered3.shape
            fin3 = bool(torch.isfinite(lg3.float()).all()) if gathered3 is None else bool(torch.isfinite(gathered3).all())
            # per-call kernel times of one more call of the same kind (rank 0; torch.profiler activity records of the eager
            # launches): the in-scope kernels of a 4096-token call over the full ring, grouped
            kern3 = None
            if rank == 0 and not args.no_kernel_timing:
                try:
                    import collections
                    from torch.profiler import ProfilerActivity, profile
                    torch.cuda.synchronize()
                    with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof3:
                        call3()
                        torch.cuda.synchronize()
                    grp = collections.defaultdict(lambda: [0.0, 0])
                    per3 = collections.defaultdict(lambda: [0.0, 0])      # the path's own kernels, by name: duration INSIDE the call
                    for ev in prof3.events():
                        if "cuda" not in str(ev.device_type).lower():
                            continue
                        nm = ev.name
                        dur = float(ev.device_time if hasattr(ev, "device_time") else ev.cuda_time)
                        if "ivl::" in nm:
                            short = nm.split("(")[0].replace("void ", "").replace("ivl::", "")
                            per3[short][0] += dur
                            per3[short][1] += 1
                        key = ("gdn_chunk (pre-pass + scan)" if "gdn_chunk" in nm else
                               "swa (rope pre-pass + prefill + combine/append)" if ("swa_" in nm) else
                               "gated norm / add+norm / SwiGLU gate / rope tables" if "ivl::" in nm else "library GEMMs and torch glue")
                        grp[key][0] += dur
                        grp[key][1] += 1
                    kern3 = {k: {"ms_per_call": round(v[0] * 1e-3, 4), "launches": v[1]} for k, v in grp.items()}
                    kern3["by_kernel_us"] = {k: {"avg_us": round(v[0] / v[1], 2), "launches": v[1]} for k, v in sorted(per3.items())}
                except Exception as e:                   # the breakdown is optional: never lose the leg over the profiler
                    kern3 = {"error": repr(e)}
            cfg3 = {"workload": f"configs[3]: bulk prefill of ONE long sequence PER GPU ({world} sequence(s), batch-sharded, no data-path "
                                f"collective) in 4096-token calls (eager launches) over a full 4096-key ring with carried GDN state; "
                                + (f"a fresh {n_timed * Tb}-token sequence per rank from position 0" if fresh else
                                   "continuing from the context the streaming + decode legs reached")
                                + "; last-position logits gathered once at the end",
                    "n_gpus": world, "sequences": world, "context_start": ctx0, "context_end": cache3.get_seq_length(), "calls_timed": n_timed,
                    "tokens_per_call": Tb, "tokens_timed_all_ranks": world * n_timed * Tb,
                    "ms_per_call": max(per_rank_ms), "ms_per_call_min": min(times3) * 1e3,
                    "ms_per_call_full_ring": sum(full) / len(full) * 1e3, "per_rank_ms": [round(v, 3) for v in per_rank_ms],
                    "region_ms_max_over_ranks": region3 * 1e3,
                    # the aggregate over all ranks: every rank's tokens over the slowest rank's barrier-to-barrier time
                    "aggregate_tok_s": world * n_timed * Tb / region3,
                    "prefill_tok_s": Tb / mean3, "per_gpu_tok_s": n_timed * Tb / region3,
                    "logits_finite": fin3,
                    "gathered_logits_shape": list(gathered3.shape) if gathered3 is not None else [1, cfg.vocab_size],
                    "gather_ms": gather3_ms if gathered3 is not None else None,
                    "mem_allocated_gib_after_ring_fill_and_at_end": [round(m / 2 ** 30, 3) for m in mem_marks],
                    "constant_memory": (len(mem_marks) < 2) or (mem_marks[-1] <= mem_marks[0]),
                    "peak_mem_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
                    "kernel_ms_in_one_call": kern3}
        del xb, cache3

    # ---- configs[1] leg (EVERY rank: one sequence per GPU; reported beside the headline, never mixed into `value`): one
    #      4096-token prefill call on a fresh cache (chunk path over 64 chunks; SWA purely causal) + 128 graphed decode steps;
    #      the slowest rank defines both times (MAX over ranks), aggregates are all ranks' tokens over them
    cfg1 = None
    if not args.no_cfg1:
        del step, dec
        Tp, reps = 4096, 3
        err1, pf, tdec, fin1 = None, None, None, None
        try:
            ids = torch.randint(0, cfg.vocab_size, (1, Tp), device=device,
                                generator=torch.Generator(device=device).manual_seed(1 + rank))
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
            fin1 = bool(torch.isfinite(lg.float()).all())
            del dec1, cache1
        except Exception as e:               # noqa: BLE001
            err1 = f"{type(e).__name__}: {e}"
        if not all_ok(err1 is None):
            cfg1 = {"error": err1 or "failed on another rank", "failed": True}
        else:
            pf, tdec = ivd.max_over_ranks(pf, device), ivd.max_over_ranks(tdec, device)
            fin1 = all_ok(fin1)
            cfg1 = {"workload": f"configs[1]: 4096-token prefill in one call (eager launches) + 128 graphed decode steps, B=1 per GPU, "
                                f"{world} GPU(s) (batch-sharded; slowest rank's times)",
                    "n_gpus": world, "prefill_ms": pf * 1e3, "prefill_tok_s": world * Tp / pf, "decode_tok_s": world * 128 / tdec,
                    "decode_ms_per_token": tdec / 128 * 1e3, "logits_finite": fin1}

    kernels, cpu = None, None
    if rank == 0 and not args.no_kernel_timing:
        kernels = kernel_timings(device, T, args.window, live_prefill=live_prefill, live_decode=live_decode)
    if rank == 0 and not args.no_cpu_baseline:          # rank 0 at any N (the other ranks wait in the barrier below)
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
            # a kernel's cost inside the step = calls per step x its duration INSIDE the real step where the committed
            # rocprofv3 trace has it (in_step_us), else the live HIP-event time over rotating inputs
            def call_ms(r):
                return r["in_step_us"] * 1e-3 if "in_step_us" in r else r["ms"]
            prefill_kernels = {k: v for k, v in kernels.items() if v["phase"] == "prefill"}
            decode_kernels = {k: v for k, v in kernels.items() if v["phase"] == "decode"}
            dom = max(prefill_kernels, key=lambda k: call_ms(prefill_kernels[k]) * prefill_kernels[k]["launches_per_step"])
            r = kernels[dom]
            tr = pmc_traffic(dom, T, args.window)
            work = r.get("alg_bytes", r.get("alg_flops"))
            # `achieved` / `frac` come from the kernel's duration INSIDE the timed workload's step (live torch.profiler
            # records of the replayed graph; falls back to the HIP-event micro-benchmark over rotating inputs).  The
            # micro-benchmark figure and the committed rocprofv3 trace are reported beside it; all three must agree.
            in_us = r.get("in_step_us")
            dur_ms = in_us * 1e-3 if in_us is not None else r["ms"]
            ach = work / (dur_ms * 1e-3) / (1e9 if r["bound"] == "hbm" else 1e12)
            out["roofline"] = {"kernel": dom, "bound": r["bound"], "achieved": ach, "peak": r["peak"],
                               "unit": r["unit"], "frac": ach / r["peak"], "traffic": tr["hbm_bytes"] if tr else None,
                               "traffic_source": tr["source"] if tr else None,
                               "algorithmic_per_launch": work, "avg_launch_ms": dur_ms,
                               "duration_source": r.get("in_step_source", "HIP events, graph of launches over rotating inputs"),
                               "launches_per_step": r["launches_per_step"],
                               "microbench_launch_ms": r["ms"], "microbench_frac": r["frac"],
                               "rocprofv3_in_step_us": r.get("in_step_us_rocprofv3"),
                               "rocprofv3_source": r.get("in_step_rocprofv3_source")}
            if r["bound"] == "hbm" and "gdn_chunk" in dom:
                # SURVEY.md 8(d)'s own byte count for the GDN call -- 24,672 B per token + 4 MiB of fp32 state read + written --
                # beside the count above, which takes the state in the dtype the cache actually holds (bf16: 2 MiB)
                survey_bytes = 24672 * T + 4 * 1024 * 1024
                out["roofline"]["survey_bytes_per_launch"] = survey_bytes
                out["roofline"]["frac_survey_bytes"] = survey_bytes / (dur_ms * 1e-3) / 1e9 / r["peak"]
            # the long-call entries at their durations INSIDE the configs[3] call (4096 tokens over a full ring, behind a 101 MB
            # GEMM output: cold inputs), beside the hot micro-benchmark figure
            by3 = ((cfg3 or {}).get("kernel_ms_in_one_call") or {}).get("by_kernel_us") or {}

            def in_call(prefixes):
                hit = [v for k, v in by3.items() if any(k.startswith(p_) for p_ in prefixes)]
                return (sum(v["avg_us"] * v["launches"] for v in hit), max(v["launches"] for v in hit)) if hit else (None, 0)
            for key, prefixes in (("gdn_chunk_fused@T=4096", ("gdn_chunk_single_kernel", "gdn_chunk_prepare_kernel", "gdn_chunk_scan_kernel")),
                                  # (the configs[3] calls over a full ring run the 256-row kernel since round 6; with IVL_SWA_RING256=0 the
                                  #  128-row entry gets the in-call figure instead)
                                  ("swa_ring256@T=4096(full ring, rope + append)", ("swa_linearize_kernel", "swa_ring256_kernel")),
                                  ("swa_prefill@T=4096(full ring, rope + append)", ("swa_rope_prepass_kernel", "swa_prefill_kernel",
                                                                                     "swa_cache_append_kernel", "swa_combine_kernel"))):
                tot, nl = in_call(prefixes)
                if tot is not None and key in kernels and nl:
                    layers_ = 27 if key.startswith("gdn") else 9
                    us = tot / layers_
                    r_ = kernels[key]
                    work_ = r_.get("alg_bytes", r_.get("alg_flops"))
                    kernels[key]["in_call_us"] = us
                    kernels[key]["in_call_source"] = "torch.profiler records of one configs[3] call (cfg3_512k_prefill.kernel_ms_in_one_call.by_kernel_us), per layer"
                    kernels[key]["frac_in_call"] = work_ / (us * 1e-6) / (1e9 if r_["bound"] == "hbm" else 1e12) / r_["peak"]
            out["kernels"] = {k: {kk: (round(vv, 6) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                              for k, v in kernels.items()}
            out["hot_path_ms_per_step"] = sum(call_ms(v) * v["launches_per_step"] for v in prefill_kernels.values())
            out["hot_path_ms_per_step_microbench"] = sum(v["ms"] * v["launches_per_step"] for v in prefill_kernels.values())
            # what is NOT the path's kernels: library GEMMs at M = 256 (stock hipBLASLt, out of SURVEY.md section 8's scope) + torch glue
            out["gemm_ms_per_step"] = out["ms_per_step"] - out["hot_path_ms_per_step"]
            out["in_scope_share_of_step"] = out["hot_path_ms_per_step"] / out["ms_per_step"]
            out["hot_path_ms_per_decode_token"] = sum(call_ms(v) * v["launches_per_decode_token"]
                                                      for k, v in decode_kernels.items() if not k.startswith("decode linear"))
            out["decode_linear_ms_per_token"] = sum(call_ms(v) * v["launches_per_decode_token"]
                                                    for k, v in decode_kernels.items() if k.startswith("decode linear"))
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if fp8 is not None:
            out["fp8_e4m3"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in fp8.items()}
        if cfg3 is not None:
            out["cfg3_512k_prefill"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cfg3.items()}
        if dist_info is not None:
            if sp is not None:
                dist_info["sp"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sp.items()}
            out["dist"] = dist_info
        if cfg1 is not None:
            out["cfg1_4k_prefill_decode"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cfg1.items()}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
