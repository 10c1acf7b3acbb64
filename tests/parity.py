"""Parity helpers: run the HIP path (through the C ABI) and the CPU oracle on the same seeded
inputs and return error metrics.  Used by the -m gpu tests, __graft_entry__.smoke() and tests/gpu_diag.py.
TEST INFRASTRUCTURE: lives under tests/ because it imports `oracle` (only tests/, smoke() and bench.py's cpu_baseline
leg may)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from oracle import gdn as ogdn
from oracle import model as omodel
from oracle import swa as oswa


def rms_rel(ref: torch.Tensor, got: torch.Tensor) -> float:
    """fla's convention (fla:ops/utils/testing.py:12-16): RMS(err) / RMS(ref)."""
    ref, got = ref.detach().double().flatten().cpu(), got.detach().double().flatten().cpu()
    return float((ref - got).square().mean().sqrt() / (ref.square().mean().sqrt() + 1e-12))


def max_abs(ref: torch.Tensor, got: torch.Tensor) -> float:
    return float((ref.detach().double().cpu() - got.detach().double().cpu()).abs().max())


def gdn_inputs(seed: int, B: int, T: int, H: int, K: int = 128, V: int = 256, with_h0: bool = True):
    """Distributions of the fla docstrings (chunk.py:325-349); values snapped to the bf16 grid."""
    g_ = torch.Generator().manual_seed(seed)
    sn = lambda x: x.to(torch.bfloat16).float()  # noqa: E731
    q = sn(torch.randn(B, T, H, K, generator=g_))
    k = sn(torch.randn(B, T, H, K, generator=g_))
    v = sn(torch.randn(B, T, H, V, generator=g_))
    beta = sn(torch.randn(B, T, H, generator=g_).sigmoid())
    g = F.logsigmoid(torch.randn(B, T, H, generator=g_))
    h0 = sn(torch.randn(B, H, K, V, generator=g_)) if with_h0 else None
    return q, k, v, g, beta, h0


def gdn_op_parity(device: str, mode: str, B: int, T: int, H: int, seed: int = 0, with_h0: bool = True,
                  state_dtype: torch.dtype = torch.float32, inplace_state: bool = False, mma_dtype=None) -> Dict[str, float]:
    """HIP chunk / recurrent op vs the oracle.  Returns errors vs the exact fp32 oracle
    (`*_vs_exact`) and vs the oracle with the reference's bf16 rounding points (`*_vs_bf16model`)."""
    from infinitevl_amd import ops
    q, k, v, g, beta, h0 = gdn_inputs(seed, B, T, H, with_h0=with_h0)
    if h0 is not None and state_dtype == torch.bfloat16:
        h0 = h0.to(torch.bfloat16).float()
    dev = torch.device(device)
    qd, kd, vd, bd = (x.to(dev, torch.bfloat16) for x in (q, k, v, beta))
    gd = g.to(dev)
    h0d = h0.to(dev, state_dtype) if h0 is not None else None
    fn = ops.chunk_gated_delta_rule if mode == "chunk" else ops.fused_recurrent_gated_delta_rule
    if mma_dtype is not None:
        import functools
        fn = functools.partial(fn, mma_dtype=mma_dtype)
    if inplace_state:
        out_state = h0d if h0d is not None else torch.zeros(B, H, 128, 256, dtype=state_dtype, device=dev)
        o, ht = fn(qd, kd, vd, gd, bd, initial_state=h0d, use_qk_l2norm_in_kernel=True, final_state_out=out_state)
    else:
        o, ht = fn(qd, kd, vd, gd, bd, initial_state=h0d, output_final_state=True, use_qk_l2norm_in_kernel=True)
    torch.cuda.synchronize()
    o_ex, s_ex = ogdn.gdn_recurrent(q, k, v, g, beta, initial_state=h0) if T <= 256 else \
        ogdn.gdn_chunk(q, k, v, g, beta, initial_state=h0)
    res = {"o_vs_exact": rms_rel(o_ex, o.float()), "s_vs_exact": rms_rel(s_ex, ht.float())}
    if mode == "chunk":
        o_bf, s_bf = ogdn.gdn_chunk(q, k, v, g, beta, initial_state=h0, rounding=torch.bfloat16)
    else:
        o_bf, s_bf = ogdn.gdn_recurrent(q, k, v, g, beta, initial_state=h0, qk_round_dtype=torch.bfloat16)
    res["o_vs_bf16model"] = rms_rel(o_bf.to(torch.bfloat16).float(), o.float())
    res["s_vs_bf16model"] = rms_rel(s_bf, ht.float())
    res["o_bf16model_vs_exact"] = rms_rel(o_ex, o_bf.to(torch.bfloat16).float())      # the reference-rounding model's own distance
    res["s_bf16model_vs_exact"] = rms_rel(s_ex, s_bf)
    if mma_dtype is not None and mode == "chunk":
        o_f8, s_f8 = ogdn.gdn_chunk(q, k, v, g, beta, initial_state=h0, rounding=torch.bfloat16, mma_rounding=torch.float8_e4m3fn)
        res["o_vs_fp8model"] = rms_rel(o_f8.to(torch.bfloat16).float(), o.float())
        res["s_vs_fp8model"] = rms_rel(s_f8, ht.float())
    res["finite"] = float(torch.isfinite(o.float()).all() and torch.isfinite(ht.float()).all())
    return res


def swa_op_parity(device: str, B: int, T: int, Hq: int, Hkv: int, window: int, seen: int, seed: int = 0,
                  via: str = "ring", d: int = 128, mma_dtype=None) -> Dict[str, float]:
    """HIP SWA vs oracle.  `seen` tokens precede the call.  via="ring": the previous keys are fed
    through the ring-buffer cache (filled by earlier appends); via="cat": operator-level call with the
    concatenated K/V (swa_attention_interface)."""
    from infinitevl_amd import ops
    g_ = torch.Generator().manual_seed(seed)
    sn = lambda x: x.to(torch.bfloat16).float()  # noqa: E731
    total = seen + T
    q = sn(torch.randn(B, T, Hq, d, generator=g_))
    k_all = sn(torch.randn(B, total, Hkv, d, generator=g_))
    v_all = sn(torch.randn(B, total, Hkv, d, generator=g_))
    n_prev = oswa.n_prev_keys(window, seen)
    ref = oswa.swa_attention(q.transpose(1, 2), k_all[:, seen - n_prev:].transpose(1, 2),
                             v_all[:, seen - n_prev:].transpose(1, 2), n_prev, window, d ** -0.5)
    dev = torch.device(device)
    qd = q.to(dev, torch.bfloat16)
    kd, vd = k_all.to(dev, torch.bfloat16), v_all.to(dev, torch.bfloat16)
    if via == "cat":
        out, _ = ops.swa_attention_interface(None, qd.transpose(1, 2), kd[:, seen - n_prev:].transpose(1, 2),
                                             vd[:, seen - n_prev:].transpose(1, 2), None, scaling=d ** -0.5,
                                             sliding_window=window)
    else:
        C = window - 1
        kc = torch.zeros(B, Hkv, C, d, dtype=torch.bfloat16, device=dev)
        vc = torch.zeros_like(kc)
        pos_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        # feed the history in irregular pieces to exercise wrap-around
        pos, step = 0, max(1, min(seen, 37))
        while pos < seen:
            n = min(step, seen - pos)
            ops.swa_cache_append(kd[:, pos:pos + n], vd[:, pos:pos + n], kc, vc, pos_dev=pos_dev)
            ops.counter_add(pos_dev, n)
            pos += n
            step = step * 2 + 1
        out = ops.swa_forward(qd, kd[:, seen:], vd[:, seen:], window=window, scaling=d ** -0.5,
                              k_cache=kc, v_cache=vc, pos_dev=pos_dev, mma_dtype=mma_dtype)
    torch.cuda.synchronize()
    res = {"o": rms_rel(ref, out.float()), "max_abs": max_abs(ref, out.float()),
           "finite": float(torch.isfinite(out.float()).all())}
    if mma_dtype is not None:
        ref8 = oswa.swa_attention(q.transpose(1, 2), k_all[:, seen - n_prev:].transpose(1, 2),
                                  v_all[:, seen - n_prev:].transpose(1, 2), n_prev, window, d ** -0.5,
                                  mma_rounding=torch.float8_e4m3fn)
        res["o_vs_fp8model"] = rms_rel(ref8, out.float())
        res["fp8model_vs_exact"] = rms_rel(ref, ref8)
    return res


# ---------------------------------------------------------------------------------------------
# layer / stack level
# ---------------------------------------------------------------------------------------------
def small_configs(window: int = 96, n_layers: int = 4, heads: int = 2):
    """Real head shapes (K=128, V=256, d=128: what the kernels are built for) with few heads."""
    from infinitevl_amd.harness import InfiniteVLTextConfig
    hidden = heads * 128
    lt = ["sliding_attention" if i % 4 == 0 else "linear_attention" for i in range(n_layers)]
    hc = InfiniteVLTextConfig(vocab_size=512, hidden_size=hidden, intermediate_size=2 * hidden,
                              num_hidden_layers=n_layers, num_attention_heads=heads, num_key_value_heads=1,
                              head_dim=128, sliding_window=window, layer_types=lt, num_linear_heads=heads,
                              num_linear_key_value_heads=heads, linear_head_dim=128, rope_theta=1e6)
    oc = omodel.OracleConfig(hidden_size=hidden, intermediate_size=2 * hidden, num_attention_heads=heads,
                             num_key_value_heads=1, num_linear_heads=heads, linear_head_dim=128, expand_v=2.0,
                             conv_size=4, sliding_window=window, rope_theta=1e6, mrope_section=[16, 24, 24],
                             layer_types=lt)
    return hc, oc


def load_params(stack: torch.nn.Module, params: Dict[str, torch.Tensor]) -> None:
    sd = stack.state_dict()
    for k, v in params.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    missing = [k for k in sd if k not in params and "inv_freq" not in k]
    assert not missing, missing
    stack.load_state_dict({k: v.clone() for k, v in params.items()}, strict=False)


def bf16_params(params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Snap weights to the bf16 grid (A_log / dt_bias included: the module is cast to bf16 as a whole)."""
    return {k: v.to(torch.bfloat16).float() for k, v in params.items()}


def layer_parity(device: str = "cuda:0", T_prefill: int = 130, n_decode: int = 3, window: int = 96,
                 seed: int = 0, stream_T: int = 70, fuse: bool = False, heads: int = 2,
                 noise_floor: bool = False, schedule=None) -> Dict[str, float]:
    """4-layer stack (1 SWA + 3 GDN), real head dims: prefill (chunk path) -> streaming frame (chunk path,
    carry-in conv, ring wrap) -> decode steps (recurrent path), HIP modules vs oracle with bf16 activations."""
    from infinitevl_amd.harness import InfiniteVLTextStack
    hc, oc = small_configs(window, heads=heads)
    params = bf16_params(omodel.random_params(oc, seed=seed, vocab=hc.vocab_size))
    stack = InfiniteVLTextStack(hc)
    load_params(stack, params)
    stack = stack.to(device=device, dtype=torch.bfloat16).eval()
    if fuse:
        stack.fuse_()          # fused projections + prologue/epilogue kernels
    cache = stack.allocate_inference_cache(1)
    ocache = omodel.new_cache(oc, cache_dtype=torch.bfloat16)
    # noise_floor: a third run, the oracle with fp32 activations / no kernel rounding, to express both bf16 results
    # (ours and the oracle's bf16 model) as distances from the exact arithmetic
    xcache = omodel.new_cache(oc, cache_dtype=torch.float32) if noise_floor else None
    g_ = torch.Generator().manual_seed(seed + 1)
    res: Dict[str, float] = {}
    pos = 0
    with torch.no_grad():
        calls = [("prefill", T_prefill), ("stream", stream_T)] + [(f"decode{i}", 1) for i in range(n_decode)]
        if schedule is not None:                      # an arbitrary sequence of call lengths over one cache
            calls = [(f"call{i}", int(T)) for i, T in enumerate(schedule)]
        for name, T in calls:
            x = (torch.randn(1, T, hc.hidden_size, generator=g_) * 0.5).to(torch.bfloat16).float()
            pid = torch.arange(pos, pos + T)[None, None, :].expand(3, 1, T).contiguous()
            h_ref = omodel.text_stack(params, x, pid, oc, ocache, act_dtype=torch.bfloat16,
                                      kernel_rounding=torch.bfloat16)
            h, _ = stack(inputs_embeds=x.to(device, torch.bfloat16), position_ids=pid.to(device),
                         past_key_values=cache, logits_to_keep=0)
            torch.cuda.synchronize()
            res[name] = rms_rel(h_ref, h.float())
            if noise_floor:
                h_ex = omodel.text_stack(params, x, pid, oc, xcache, act_dtype=torch.float32, kernel_rounding=None)
                res[name + "_vs_exact"] = rms_rel(h_ex, h.float())
                res[name + "_bf16model_vs_exact"] = rms_rel(h_ex, h_ref)
            pos += T
    res["gdn_state"] = rms_rel(ocache[1].recurrent, cache.layers[1].recurrent_state.float())
    res["swa_keys"] = rms_rel(ocache[0].k, cache.layers[0].keys.float())
    return res


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[0] / SURVEY.md 8d cfg1: the single-image greedy-decode plumbing run
# ---------------------------------------------------------------------------------------------
def configs0_workload(hidden: int, vocab: int, seed: int = 0, n_text: int = 64, grid: int = 16):
    """The shape of data/mllm_demo.json sample 0 after the processor (std:2112-2176; the ViT and the tokenizer are
    out of scope and there is no checkpoint offline): `n_text` random token ids (seed 0) around a grid x grid block of
    image-placeholder positions whose embeddings are replaced by randn * 0.02 "vision features".  Layout: 8 text tokens,
    the image block, the remaining text.  position_ids are the 3-D (t, h, w) M-RoPE positions of Qwen2.5-VL's
    get_rope_index: text advances all three together, the image block keeps t fixed and counts rows / columns.
    Returns (ids [1,n_text], image_embeds [1,grid*grid,hidden] fp32 on the bf16 grid, is_image [T] bool, position_ids
    [3,1,T] int64, next_position)."""
    g_ = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab, (1, n_text), generator=g_)
    img = (torch.randn(1, grid * grid, hidden, generator=g_) * 0.02).to(torch.bfloat16).float()
    n_img, pre = grid * grid, 8
    T = n_text + n_img
    is_image = torch.zeros(T, dtype=torch.bool)
    is_image[pre:pre + n_img] = True
    pos = torch.zeros(3, T, dtype=torch.int64)
    pos[:, :pre] = torch.arange(pre)
    rr, cc = torch.meshgrid(torch.arange(grid), torch.arange(grid), indexing="ij")
    pos[0, pre:pre + n_img] = pre
    pos[1, pre:pre + n_img] = pre + rr.flatten()
    pos[2, pre:pre + n_img] = pre + cc.flatten()
    nxt = pre + grid
    pos[:, pre + n_img:] = nxt + torch.arange(T - pre - n_img)
    return ids, img, is_image, pos[:, None, :].contiguous(), nxt + (T - pre - n_img)


def configs0_embeds(embed: torch.Tensor, ids: torch.Tensor, img: torch.Tensor, is_image: torch.Tensor) -> torch.Tensor:
    """inputs_embeds with the image features spliced over the placeholder positions (std:2140-2160)."""
    T = is_image.numel()
    x = torch.zeros(1, T, embed.shape[1], dtype=embed.dtype, device=embed.device)
    x[0, ~is_image.to(embed.device)] = embed[ids[0].to(embed.device)]
    x[0, is_image.to(embed.device)] = img[0].to(embed.device, embed.dtype)
    return x


def configs0_oracle_run(oc, params, n_new: int = 16, act_dtype=None, kernel_rounding=None, forced_tokens=None):
    """Prefill + greedy decode of `n_new` tokens on the CPU oracle.  Returns (tokens, logits [n_new, vocab], cache)."""
    embed = params["embed_tokens.weight"]
    ids, img, is_image, pos3, nxt = configs0_workload(oc.hidden_size, embed.shape[0])
    cache = omodel.new_cache(oc, cache_dtype=act_dtype)
    x = configs0_embeds(embed, ids, img, is_image)
    h = omodel.text_stack(params, x, pos3, oc, cache, act_dtype=act_dtype, kernel_rounding=kernel_rounding)
    toks, logits = [], []
    lg = h[0, -1].float() @ embed.float().T
    for step in range(n_new):
        logits.append(lg)
        tok = int(lg.argmax())
        toks.append(tok)
        feed = tok if forced_tokens is None else int(forced_tokens[step])
        pid = torch.full((3, 1, 1), nxt + step, dtype=torch.int64)
        h = omodel.text_stack(params, embed[torch.tensor([[feed]])], pid, oc, cache, act_dtype=act_dtype,
                              kernel_rounding=kernel_rounding)
        lg = h[0, -1].float() @ embed.float().T
    return toks, torch.stack(logits), cache
