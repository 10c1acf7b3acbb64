#!/usr/bin/env python3
"""Generate the committed golden fixtures by EXECUTING THE REFERENCE in the build container.

    python tests/golden/gen_golden.py            # (re)writes tests/golden/*.npz

Requires /root/reference (read-only) and no GPU: the vendored flash-linear-attention
Triton kernels run under TRITON_INTERPRET=1 (see _ref_loader.py for the recipe and the
list of shims, all applied on OUR side).  The fixtures are pure data: inputs + the
reference's outputs.  The tests never import the reference.

Conventions
  * Operator inputs are drawn in fp32 and snapped to the bf16 grid, then handed to the
    reference as fp32 tensors (numpy has no bf16, so the interpreter cannot run bf16).
    They are stored as uint16 bf16 bit patterns (`*_bf16bits`): the same numbers can be
    fed to the HIP kernels as genuine bf16 with no input rounding.
  * Distributions follow the fla docstrings (chunk.py:325-349): q,v~randn,
    k~randn (l2norm in kernel), beta~sigmoid(randn), g~logsigmoid(randn), h0~randn.
"""
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

from _ref_loader import load_reference_fla, load_reference_std  # noqa: E402

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def snap(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).float()


def bits(x: torch.Tensor) -> np.ndarray:
    return x.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ---------------------------------------------------------------------------
def gdn_inputs(seed, B, T, H, K, V, with_h0):
    g_ = torch.Generator().manual_seed(seed)
    q = snap(torch.randn(B, T, H, K, generator=g_))
    k = snap(torch.randn(B, T, H, K, generator=g_))
    v = snap(torch.randn(B, T, H, V, generator=g_))
    beta = snap(torch.randn(B, T, H, generator=g_).sigmoid())
    g = F.logsigmoid(torch.randn(B, T, H, generator=g_))        # fp32 (g stays fp32 in the model, std:1294)
    h0 = snap(torch.randn(B, H, K, V, generator=g_)) if with_h0 else None
    return q, k, v, g, beta, h0


def gen_gdn_ops(ns):
    print("[gdn ops] vendored fused_recurrent / chunk kernels under the Triton interpreter")
    cases = [
        # name,           mode,        seed, B, T,   H, K,   V,   h0
        ("rec_T1_h0",     "recurrent", 0,    2, 1,   2, 128, 256, True),
        ("rec_T7_h0",     "recurrent", 1,    1, 7,   2, 128, 256, True),
        ("rec_T64",       "recurrent", 2,    1, 64,  2, 128, 256, False),
        ("chunk_T65_h0",  "chunk",     3,    1, 65,  2, 128, 256, True),
        ("chunk_T160",    "chunk",     0,    1, 160, 2, 128, 256, False),
        ("chunk_T256_h0", "chunk",     1,    2, 256, 2, 128, 256, True),
        ("chunk_T100_small", "chunk",  2,    1, 100, 3, 32,  64,  True),
    ]
    for name, mode, seed, B, T, H, K, V, with_h0 in cases:
        t0 = time.time()
        q, k, v, g, beta, h0 = gdn_inputs(seed, B, T, H, K, V, with_h0)
        fn = ns.recurrent_tm if mode == "recurrent" else ns.chunk_tm
        o, ht = fn(q, k, v, g, beta, initial_state=h0, output_final_state=True, use_qk_l2norm_in_kernel=True)
        arrs = dict(q_bf16bits=bits(q), k_bf16bits=bits(k), v_bf16bits=bits(v), beta_bf16bits=bits(beta),
                    g=g, o=o.float(), ht=ht.float(), mode=np.array(mode))
        if h0 is not None:
            arrs["h0_bf16bits"] = bits(h0)
        save("gdn_" + name, **arrs)
        print(f"    {name}: {time.time() - t0:.1f}s")

    # chained: 2 x 128 tokens with the carried state rounded to bf16 in between
    q, k, v, g, beta, h0 = gdn_inputs(3, 1, 256, 2, 128, 256, True)
    o1, s1 = ns.chunk_tm(q[:, :128], k[:, :128], v[:, :128], g[:, :128], beta[:, :128], initial_state=h0,
                         output_final_state=True, use_qk_l2norm_in_kernel=True)
    s1r = snap(s1)
    o2, s2 = ns.chunk_tm(q[:, 128:], k[:, 128:], v[:, 128:], g[:, 128:], beta[:, 128:], initial_state=s1r,
                         output_final_state=True, use_qk_l2norm_in_kernel=True)
    save("gdn_chunk_chained_2x128_bf16state", q_bf16bits=bits(q), k_bf16bits=bits(k), v_bf16bits=bits(v),
         beta_bf16bits=bits(beta), g=g, h0_bf16bits=bits(h0), o=torch.cat([o1, o2], 1).float(),
         s_mid_bf16bits=bits(s1r), ht=s2.float(), mode=np.array("chunk"))

    # fp16 through the PUBLIC wrapper (pins the reduced-precision rounding points)
    q, k, v, g, beta, h0 = gdn_inputs(0, 1, 160, 2, 128, 256, True)
    qh, kh, vh, bh = (x.half() for x in (q, k, v, beta))
    o, ht = ns.chunk_tm(qh, kh, vh, g, bh, initial_state=h0, output_final_state=True, use_qk_l2norm_in_kernel=True)
    save("gdn_chunk_T160_fp16", q_f16=qh, k_f16=kh, v_f16=vh, beta_f16=bh, g=g, h0_bf16bits=bits(h0),
         o=o.float(), ht=ht.float(), mode=np.array("chunk"))


def gen_conv_norm(ns):
    print("[short conv / gated rmsnorm] vendored modules")
    from einops import rearrange
    torch.manual_seed(0)
    D, W = 64, 4
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        conv = ns.conv.ShortConvolution(D, W, activation="silu", use_fast_conv1d=False)
    with torch.no_grad():
        conv.weight.copy_(snap(torch.randn(D, 1, W) * 0.5))
    x = snap(torch.randn(2, 10, D))
    with torch.no_grad():
        y, cache = conv(x, cache=None, output_final_state=True)
        ys, caches = [], []
        xs = snap(torch.randn(5, 2, 1, D))
        c = cache.clone()
        for i in range(5):
            yi, c = conv(xs[i], cache=c, output_final_state=True)
            ys.append(yi.clone())
            caches.append(c.clone())
        x3 = snap(torch.randn(2, 3, D))                     # T < W prefill
        y3, cache3 = conv(x3, cache=None, output_final_state=True)
    save("short_conv", weight_bf16bits=bits(conv.weight.detach()), x_bf16bits=bits(x), y=y, state=cache,
         xs_bf16bits=bits(xs), ys=torch.stack(ys), states=torch.stack(caches),
         x3_bf16bits=bits(x3), y3=y3, state3=cache3)

    torch.manual_seed(1)
    norm = ns.norm_gate.FusedRMSNormGated(256, eps=1e-5)
    with torch.no_grad():
        norm.weight.copy_(snap(1.0 + 0.1 * torch.randn(256)))
    xo = snap(torch.randn(2, 5, 4, 256) * 2.0)
    gate = snap(torch.randn(2, 5, 4, 256))
    with torch.no_grad():
        yo = norm(xo, gate)
    save("rmsnorm_gate", weight_bf16bits=bits(norm.weight.detach()), x_bf16bits=bits(xo), gate_bf16bits=bits(gate),
         y=yo, eps=np.float32(1e-5))


def gen_swa(ns):
    print("[swa] reference eager_attention_forward + HF mask predicates + rope")
    std = ns.std
    from transformers import masking_utils as mu

    # --- integer band fixtures from HF's own predicate (kv > q - W) & (kv <= q)
    rows = []
    for W in (8, 5, 4096):
        for seen in (0, 3, W - 2, W - 1, W, W + 5, 3 * W):
            for T in (1, 2, W - 1, W, W + 3):
                if T <= 0 or T > 300 and W > 100 and T not in (1, 2):
                    continue
                if W > 100 and T > 2:
                    T_ = 64
                else:
                    T_ = T
                n_prev = min(W - 1, seen)
                fn = mu.sliding_window_causal_mask_function(W)
                q_abs = torch.arange(seen, seen + T_)[:, None]
                kv_abs = torch.arange(seen - n_prev, seen + T_)[None, :]
                vis = fn(0, 0, q_abs, kv_abs)
                vis = vis.numpy()
                lo = vis.argmax(1)
                hi = vis.shape[1] - 1 - vis[:, ::-1].argmax(1)
                assert all(vis[i, lo[i]:hi[i] + 1].all() and vis[i].sum() == hi[i] - lo[i] + 1 for i in range(T_))
                rows.append((W, seen, T_, n_prev, lo.astype(np.int64), hi.astype(np.int64)))
    save("swa_band", W=np.array([r[0] for r in rows]), seen=np.array([r[1] for r in rows]),
         T=np.array([r[2] for r in rows]), n_prev=np.array([r[3] for r in rows]),
         lo=np.concatenate([r[4] for r in rows]), hi=np.concatenate([r[5] for r in rows]))

    # --- attention math through the reference's eager function with the band
    class _M:  # what eager_attention_forward reads from `module`
        num_key_value_groups = 4
        training = False

    torch.manual_seed(2)
    B, Hq, Hkv, d, W = 2, 8, 2, 64, 8
    att = {}
    for name, n_prev, T in (("empty_Tlt", 0, 5), ("empty_Tgt", 0, 19), ("cached_full", 7, 6), ("decode", 7, 1),
                            ("cached_part", 3, 9)):
        q = snap(torch.randn(B, Hq, T, d))
        k = snap(torch.randn(B, Hkv, n_prev + T, d))
        v = snap(torch.randn(B, Hkv, n_prev + T, d))
        i = torch.arange(T)[:, None] + n_prev
        j = torch.arange(n_prev + T)[None, :]
        vis = (j <= i) & (j > i - W)
        mask = torch.zeros(T, n_prev + T).masked_fill(~vis, float("-inf"))[None, None]
        out, _ = std.eager_attention_forward(_M(), q, k, v, mask, scaling=d ** -0.5)
        att.update({f"{name}_q_bf16bits": bits(q), f"{name}_k_bf16bits": bits(k), f"{name}_v_bf16bits": bits(v),
                    f"{name}_out": out, f"{name}_n_prev": np.int64(n_prev)})
    save("swa_attention", W=np.int64(W), **att)

    # --- M-RoPE: reference rotary module + apply function
    from infinitevl.infinitevl_standard.configuration_infinitevl import InfiniteVLTextConfig
    cfg = InfiniteVLTextConfig(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                               num_hidden_layers=1, rope_theta=1e6, layer_types=["sliding_attention"],
                               rope_scaling={"type": "default", "rope_type": "default", "mrope_section": [16, 24, 24]})
    rot = std.InfiniteVLRotaryEmbedding(cfg)
    torch.manual_seed(3)
    T = 12
    pos = torch.stack([torch.arange(100, 100 + T), torch.randint(0, 50, (T,)), torch.randint(0, 50, (T,))])[:, None, :]
    pos = pos.expand(3, 2, T).contiguous()
    x = torch.zeros(2, T, 256)
    cos, sin = rot(x, pos)
    q = snap(torch.randn(2, 2, T, 128))
    k = snap(torch.randn(2, 1, T, 128))
    qe, ke = std.apply_multimodal_rotary_pos_emb(q, k, cos, sin, [16, 24, 24])
    save("mrope", position_ids=pos.numpy(), cos=cos, sin=sin, q_bf16bits=bits(q), k_bf16bits=bits(k), q_out=qe,
         k_out=ke, theta=np.float64(1e6), mrope_section=np.array([16, 24, 24]))


def tiny_cfg(layer_types, W=8):
    from infinitevl.infinitevl_standard.configuration_infinitevl import InfiniteVLTextConfig
    cfg = InfiniteVLTextConfig(
        vocab_size=97, hidden_size=64, intermediate_size=96, num_hidden_layers=len(layer_types),
        num_attention_heads=4, num_key_value_heads=2, head_dim=16, rms_norm_eps=1e-6, norm_eps=1e-5,
        rope_theta=1e4, use_sliding_window=True, sliding_window=W, max_window_layers=len(layer_types),
        layer_types=list(layer_types), max_position_embeddings=4096,
        rope_scaling={"type": "default", "rope_type": "default", "mrope_section": [2, 3, 3]},
        num_linear_heads=4, num_linear_key_value_heads=4, linear_head_dim=16, expand_v=2, conv_size=4,
        pad_token_id=None,
    )
    return cfg


def gen_cache_traces(ns):
    print("[cache] integer traces of the reference cache classes")
    std = ns.std
    out = {}
    for W, steps in ((8, [5, 1, 1, 1, 6, 1, 20]), (8, [1] * 12), (8, [3, 3, 3, 3]), (4096, [257, 256, 256] + [256] * 16 + [1, 1, 7])):
        cfg = tiny_cfg(["sliding_attention"], W=W)
        layer = std.StaticSlidingWindowLayerPrealloc(config=cfg, batch_size=1, dtype=torch.float32)
        tr = []
        pos = 0
        for T in steps:
            k = torch.zeros(1, 2, T, 16)
            fk, fv = layer.update(k, k)
            kv_len, kv_off = layer.get_mask_sizes(torch.arange(pos, pos + T))
            tr.append((fk.shape[2], layer.size, layer.cumulative_length, kv_len, kv_off))
            pos += T
        key = f"W{W}_" + "_".join(map(str, steps[:4])) + f"_n{len(steps)}"
        out[key + "_steps"] = np.array(steps)
        out[key + "_trace"] = np.array(tr, dtype=np.int64)
        out[key + "_W"] = np.int64(W)
    # content check of the tail buffer for the first trace
    cfg = tiny_cfg(["sliding_attention"], W=8)
    layer = std.StaticSlidingWindowLayerPrealloc(config=cfg, batch_size=1, dtype=torch.float32)
    pos = 0
    tails = []
    for T in [5, 1, 1, 1, 6, 1, 20]:
        k = torch.arange(pos, pos + T, dtype=torch.float32)[None, None, :, None].expand(1, 2, T, 16).contiguous()
        layer.update(k, -k)
        tails.append(layer.keys[0, 0, :, 0].clone().numpy())
        pos += T
    out["W8_tail_positions"] = np.concatenate(tails)
    out["W8_tail_lengths"] = np.array([len(t) for t in tails])
    # crop (std:192-213): ops = +n: update with n tokens, c: crop(c) encoded as 1000 + c (c may be negative)
    cfg = tiny_cfg(["sliding_attention"], W=8)
    layer = std.StaticSlidingWindowLayerPrealloc(config=cfg, batch_size=1, dtype=torch.float32)
    ops_, tails, ctr, pos = [3, 1000 + 2, 2, 1000 - 1, 1, 1000 + 9, 2], [], [], 0
    for op in ops_:
        if op < 500:
            k = torch.arange(pos, pos + op, dtype=torch.float32)[None, None, :, None].expand(1, 2, op, 16).contiguous()
            layer.update(k, -k)
            pos += op
        else:
            layer.crop(op - 1000)
        tails.append(layer.keys[0, 0, :, 0].clone().numpy())
        ctr.append((layer.size, layer.cumulative_length))
    out["crop_ops"] = np.array(ops_)
    out["crop_tail_positions"] = np.concatenate(tails)
    out["crop_tail_lengths"] = np.array([len(t) for t in tails])
    out["crop_counters"] = np.array(ctr, dtype=np.int64)
    save("cache_traces", **out)


def state_dict_np(mod):
    return {k: v.detach().float().numpy() for k, v in mod.state_dict().items()}


def gen_layers(ns):
    print("[layers] reference GatedDeltaNet / InfiniteVLSelfAttention / DecoderLayer on a tiny config (fp32, CPU)")
    std = ns.std
    lt = ["sliding_attention", "linear_attention", "linear_attention", "linear_attention"]
    cfg = tiny_cfg(lt, W=8)

    def build_layers(seed):
        torch.manual_seed(seed)
        layers = []
        for i in range(len(lt)):
            layer = std.InfiniteVLDecoderLayer(cfg, i)
            for n, p_ in layer.named_parameters():
                with torch.no_grad():
                    if n.endswith("A_log") or n.endswith("dt_bias"):
                        continue
                    if "layernorm" in n or "o_norm" in n:
                        p_.copy_(1.0 + 0.1 * torch.randn_like(p_))
                    elif "conv1d" in n:
                        p_.copy_(0.4 * torch.randn_like(p_))
                    elif n.endswith("bias"):
                        p_.copy_(0.1 * torch.randn_like(p_))
                    else:
                        p_.copy_(0.15 * torch.randn_like(p_))
            layer.eval()
            layers.append(layer)
        cfg._attn_implementation = "ivl_band"          # std:1028 overwrote it during construction
        return layers

    layers = build_layers(0)
    rot = std.InfiniteVLRotaryEmbedding(cfg)
    torch.manual_seed(1)
    final_norm_w = 1.0 + 0.1 * torch.randn(64)
    embed = 0.5 * torch.randn(97, 64)

    def run_stack(x, pos3, cache, start):
        T = x.shape[1]
        cache_position = torch.arange(start, start + T)
        pe = rot(x, pos3)
        per_layer = []
        for layer in layers:
            x = layer(x, attention_mask=None, position_ids=pos3, past_key_values=cache, use_cache=True,
                      cache_position=cache_position, position_embeddings=pe)[0]
            per_layer.append(x.clone())
        xf = x.float()
        h = final_norm_w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + cfg.rms_norm_eps))
        return h, per_layer

    def pos_for(start, T, B=1):
        return torch.arange(start, start + T)[None, None, :].expand(3, B, T).contiguous()

    out = {}
    for i, layer in enumerate(layers):
        for k, v in state_dict_np(layer).items():
            out[f"w.layers.{i}.{k}"] = v
    out["w.norm.weight"] = final_norm_w.numpy()
    out["w.embed_tokens.weight"] = embed.numpy()
    out["layer_types"] = np.array(lt)

    # --- scenario: prefill T=70 (chunk path), stream 2 x T=20 (recurrent path, carry-in conv),
    #     clone cache, greedy-decode 6 tokens on the clone, then one more streaming step on the original.
    with torch.no_grad():
        cache = std.StaticCachePrealloc(config=cfg, batch_size=1, dtype=torch.float32)
        torch.manual_seed(5)
        ids0 = torch.randint(0, 97, (1, 70))
        x0 = embed[ids0]
        h0, pl0 = run_stack(x0, pos_for(0, 70), cache, 0)
        out["s.ids0"] = ids0.numpy()
        out["s.h0"] = h0.numpy()
        out["s.h0_layer0"] = pl0[0].numpy()
        out["s.h0_layer1"] = pl0[1].numpy()
        pos = 70
        frames = 0.5 * torch.randn(3, 1, 20, 64)
        out["s.frames"] = frames.numpy()
        hs = []
        for f in range(2):
            h, _ = run_stack(frames[f], pos_for(pos, 20), cache, pos)
            hs.append(h.numpy())
            pos += 20
        out["s.h_frames"] = np.stack(hs)

        # clone (what demo clone_inference_cache does, demo:111-160): deep copy
        import copy
        qa = copy.deepcopy(cache)
        tok = int((h[0, -1] @ embed.T).argmax())
        toks, logits = [], []
        qpos = pos
        for step in range(6):
            hq, _ = run_stack(embed[torch.tensor([[tok]])], pos_for(qpos, 1), qa, qpos)
            lg = hq[0, -1] @ embed.T
            logits.append(lg.numpy())
            tok = int(lg.argmax())
            toks.append(tok)
            qpos += 1
        out["s.decode_tokens"] = np.array(toks)
        out["s.decode_logits"] = np.stack(logits)

        h3, _ = run_stack(frames[2], pos_for(pos, 20), cache, pos)
        out["s.h_frame2_after_clone"] = h3.numpy()
        # final cache content of the original stream
        out["s.swa_keys"] = cache.layers[0].keys.numpy()
        out["s.swa_values"] = cache.layers[0].values.numpy()
        out["s.swa_size"] = np.int64(cache.layers[0].size)
        out["s.swa_cum"] = np.int64(cache.layers[0].cumulative_length)
        out["s.gdn1_recurrent"] = cache.layers[1].recurrent_state.numpy()
        out["s.gdn1_conv_q"] = cache.layers[1].conv_state_q.numpy()
        out["s.gdn1_seq_len"] = np.int64(cache.layers[1].seq_len)
    save("tiny_stack", **out)

    # --- single mixers at the module boundary (no cache): GDN T=70 chunk, SWA T=19 > W
    with torch.no_grad():
        torch.manual_seed(7)
        x = 0.7 * torch.randn(2, 70, 64)
        o_gdn, _ = layers[1].self_attn(x, past_key_values=None)
        xs = 0.7 * torch.randn(2, 19, 64)
        pos3 = torch.stack([torch.arange(19), torch.arange(19) // 3, torch.arange(19) % 5])[:, None, :].expand(3, 2, 19).contiguous()
        pe = rot(xs, pos3)
        o_swa, _ = layers[0].self_attn(xs, attention_mask=None, position_ids=pos3, past_key_values=None,
                                        cache_position=torch.arange(19), position_embeddings=pe)
    save("tiny_mixers", x_gdn=x, o_gdn=o_gdn, x_swa=xs, pos3=pos3.numpy(), o_swa=o_swa)


def gen_swa_d128(ns):
    """The reference's eager attention at the head shape the HIP kernel is built for (d = 128, GQA group 8 as in the
    model, 16/2 heads once), band mask from the S2 predicate, W smaller and larger than T: consumed DIRECTLY by the GPU
    tests (tests/test_gpu_parity.py::test_swa_reference_vectors_d128)."""
    print("[swa_d128] eager_attention_forward at d=128")
    std = ns.std
    torch.manual_seed(12)
    att, names = {}, []
    for name, Hq, Hkv, W, n_prev, T in (("empty_Tlt", 8, 1, 96, 0, 40), ("empty_Tgt", 8, 1, 96, 0, 130),
                                        ("cached_full", 8, 1, 96, 95, 70), ("decode", 8, 1, 96, 95, 1),
                                        ("cached_part", 8, 1, 96, 30, 60), ("gqa16x2_full", 16, 2, 64, 63, 33),
                                        ("wide_window", 8, 1, 4096, 100, 29)):
        class _M:
            num_key_value_groups = Hq // Hkv
            training = False
        q = snap(torch.randn(1, Hq, T, 128))
        k = snap(torch.randn(1, Hkv, n_prev + T, 128))
        v = snap(torch.randn(1, Hkv, n_prev + T, 128))
        i = torch.arange(T)[:, None] + n_prev
        j = torch.arange(n_prev + T)[None, :]
        vis = (j <= i) & (j > i - W)
        mask = torch.zeros(T, n_prev + T).masked_fill(~vis, float("-inf"))[None, None]
        out, _ = std.eager_attention_forward(_M(), q, k, v, mask, scaling=128 ** -0.5)
        names.append(name)
        att.update({f"{name}_q_bf16bits": bits(q), f"{name}_k_bf16bits": bits(k), f"{name}_v_bf16bits": bits(v),
                    f"{name}_out": out.to(torch.float32), f"{name}_n_prev": np.int64(n_prev), f"{name}_W": np.int64(W)})
    save("swa_attention_d128", names=np.array(names), **att)


def gen_stack_realdims(ns):
    """InfiniteVLDecoderLayer x 2 (1 sliding-window + 1 Gated DeltaNet) at the head shapes the HIP kernels are built
    for: head_dim 128, linear_head_dim 128, expand_v 2 (K = 128, V = 256), mrope_section [16, 24, 24], window 96.
    Weights and inputs are snapped to the bf16 grid, so the bf16 HIP stack runs on exactly these numbers; the reference
    computes in fp32.  Same scenario as tiny_stack (prefill 70 -> 2 frames of 20 (window wraps) -> clone -> greedy
    decode 6 -> one more frame).  Consumed DIRECTLY by tests/test_gpu_parity.py::test_stack_realdims_reference_vectors."""
    print("[stack_realdims] reference decoder layers at K=128 / V=256 / d=128 (fp32, CPU, Triton interpreter)")
    std = ns.std
    from infinitevl.infinitevl_standard.configuration_infinitevl import InfiniteVLTextConfig
    lt = ["sliding_attention", "linear_attention"]
    cfg = InfiniteVLTextConfig(
        vocab_size=97, hidden_size=256, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
        num_key_value_heads=1, head_dim=128, rms_norm_eps=1e-6, norm_eps=1e-5, rope_theta=1e6, use_sliding_window=True,
        sliding_window=96, max_window_layers=2, layer_types=list(lt), max_position_embeddings=4096,
        rope_scaling={"type": "default", "rope_type": "default", "mrope_section": [16, 24, 24]},
        num_linear_heads=2, num_linear_key_value_heads=2, linear_head_dim=128, expand_v=2, conv_size=4, pad_token_id=None)
    torch.manual_seed(21)
    layers = []
    for i in range(2):
        layer = std.InfiniteVLDecoderLayer(cfg, i)
        with torch.no_grad():
            for n, p_ in layer.named_parameters():
                if n.endswith("A_log") or n.endswith("dt_bias"):
                    pass
                elif "layernorm" in n or "o_norm" in n:
                    p_.copy_(1.0 + 0.1 * torch.randn_like(p_))
                elif "conv1d" in n:
                    p_.copy_(0.4 * torch.randn_like(p_))
                elif n.endswith("bias"):
                    p_.copy_(0.1 * torch.randn_like(p_))
                else:
                    p_.copy_(torch.randn_like(p_) * (p_.shape[-1] ** -0.5))
                p_.copy_(snap(p_))
        layer.eval()
        layers.append(layer)
    cfg._attn_implementation = "ivl_band"
    rot = std.InfiniteVLRotaryEmbedding(cfg)
    final_norm_w = snap(1.0 + 0.1 * torch.randn(256))
    embed = snap(0.5 * torch.randn(97, 256))

    def run_stack(x, pos3, cache, start):
        T = x.shape[1]
        pe = rot(x, pos3)
        for layer in layers:
            x = layer(x, attention_mask=None, position_ids=pos3, past_key_values=cache, use_cache=True,
                      cache_position=torch.arange(start, start + T), position_embeddings=pe)[0]
        xf = x.float()
        return final_norm_w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + cfg.rms_norm_eps))

    def pos_for(start, T):
        return torch.arange(start, start + T)[None, None, :].expand(3, 1, T).contiguous()

    out = {"layer_types": np.array(lt)}
    for i, layer in enumerate(layers):
        for k, v in layer.state_dict().items():
            out[f"w.layers.{i}.{k}_bf16bits"] = bits(v)
    out["w.norm.weight_bf16bits"] = bits(final_norm_w)
    out["w.embed_tokens.weight_bf16bits"] = bits(embed)
    with torch.no_grad():
        cache = std.StaticCachePrealloc(config=cfg, batch_size=1, dtype=torch.float32)
        ids0 = torch.randint(0, 97, (1, 70))
        out["s.ids0"] = ids0.numpy()
        out["s.h0"] = run_stack(embed[ids0], pos_for(0, 70), cache, 0).numpy()
        pos = 70
        frames = snap(0.5 * torch.randn(3, 1, 20, 256))
        out["s.frames_bf16bits"] = bits(frames)
        hs = []
        for f in range(2):
            h = run_stack(frames[f], pos_for(pos, 20), cache, pos)
            hs.append(h.numpy())
            pos += 20
        out["s.h_frames"] = np.stack(hs)
        import copy
        qa = copy.deepcopy(cache)
        tok = int((h[0, -1] @ embed.T).argmax())
        out["s.first_token"] = np.int64(tok)
        toks, logits, qpos = [], [], pos
        for _ in range(6):
            hq = run_stack(embed[torch.tensor([[tok]])], pos_for(qpos, 1), qa, qpos)
            lg = hq[0, -1] @ embed.T
            logits.append(lg.numpy())
            tok = int(lg.argmax())
            toks.append(tok)
            qpos += 1
        out["s.decode_tokens"] = np.array(toks)
        out["s.decode_logits"] = np.stack(logits)
        out["s.h_frame2_after_clone"] = run_stack(frames[2], pos_for(pos, 20), cache, pos).numpy()
        out["s.swa_keys"] = cache.layers[0].keys.numpy()
        out["s.swa_values"] = cache.layers[0].values.numpy()
        out["s.swa_size"] = np.int64(cache.layers[0].size)
        out["s.swa_cum"] = np.int64(cache.layers[0].cumulative_length)
        out["s.gdn1_recurrent"] = cache.layers[1].recurrent_state.numpy()
        out["s.gdn1_conv_q"] = cache.layers[1].conv_state_q.numpy()
        # single mixers at the module boundary, no cache
        x = snap(0.7 * torch.randn(1, 70, 256))
        out["m.x_gdn_bf16bits"] = bits(x)
        out["m.o_gdn"] = layers[1].self_attn(x, past_key_values=None)[0].numpy()
        xs = snap(0.7 * torch.randn(1, 130, 256))
        pos3 = torch.stack([torch.arange(130), torch.arange(130) // 3, torch.arange(130) % 5])[:, None, :].contiguous()
        out["m.x_swa_bf16bits"] = bits(xs)
        out["m.pos3"] = pos3.numpy()
        out["m.o_swa"] = layers[0].self_attn(xs, attention_mask=None, position_ids=pos3, past_key_values=None,
                                             cache_position=torch.arange(130), position_embeddings=rot(xs, pos3))[0].numpy()
    save("stack_realdims", **out)


def gen_vision(ns):
    """Vision-tower window attention from the reference's own code: the vision rotary
    embedding (InfiniteVLVisionRotaryEmbedding + apply_rotary_pos_emb_vision, bf16 in / bf16 out: bit-exact pin), the
    per-window eager_attention_forward loop at head_dim 80 (the 3B tower's 1280 / 16), and the whole
    InfiniteVLVisionAttention module (eager path) at hidden 320 = 4 heads x 80."""
    print("[vision] InfiniteVLVisionAttention / apply_rotary_pos_emb_vision / eager_attention_forward at head_dim 80")
    std = ns.std
    torch.manual_seed(21)
    S, H, d = 136, 8, 80
    rot = std.InfiniteVLVisionRotaryEmbedding(d // 2)
    pos_hw = torch.randint(0, 32, (S, 2))
    rotary = rot(32)[pos_hw].flatten(1)                                   # what rot_pos_emb returns per patch
    emb = torch.cat((rotary, rotary), dim=-1)
    cos, sin = emb.cos(), emb.sin()
    out = {"pos_hw": pos_hw.numpy(), "cos": cos, "sin": sin}
    q, k, v = (torch.randn(S, H, d).to(torch.bfloat16) for _ in range(3))
    qr, kr = std.apply_rotary_pos_emb_vision(q, k, cos, sin)              # bf16 -> bf16
    out.update(q_bf16bits=bits(q), k_bf16bits=bits(k), v_bf16bits=bits(v), q_rot_bf16bits=bits(qr), k_rot_bf16bits=bits(kr))

    class _M:
        num_key_value_groups = 1
        training = False
    for name, cu in (("windows", [0, 64, 104, 136]), ("full", [0, 136]), ("ragged", [0, 1, 70, 70, 136])):
        qs, ks, vs = (x.float().transpose(0, 1).unsqueeze(0) for x in (qr, kr, v))      # [1, H, S, d] as in strm:745-747
        outs = [std.eager_attention_forward(_M(), qs[:, :, a:b], ks[:, :, a:b], vs[:, :, a:b], None, scaling=d ** -0.5)[0]
                for a, b in zip(cu[:-1], cu[1:]) if b > a]
        out[f"{name}_cu"] = np.asarray(cu, dtype=np.int32)
        out[f"{name}_out"] = torch.cat(outs, dim=1)[0]                   # [S, H, d] fp32

    # the module: hidden 320 = 4 heads x 80, eager path (the per-window loop of std:641-664)
    cfg = std.InfiniteVLVisionConfig(hidden_size=320, num_heads=4, depth=1, intermediate_size=64, out_hidden_size=64)
    cfg._attn_implementation = "eager"
    mod = std.InfiniteVLVisionAttention(cfg).eval()
    with torch.no_grad():
        for prm in mod.parameters():
            prm.copy_(snap(prm * 2.0))
    x = snap(torch.randn(S, 320))
    for kname, prm in mod.state_dict().items():
        out["mod_" + kname.replace(".", "_") + "_bf16bits"] = bits(prm)
    out["mod_x_bf16bits"] = bits(x)
    with torch.no_grad():
        for name, cu in (("windows", [0, 64, 104, 136]), ("full", [0, 136])):
            out[f"mod_{name}_out"] = mod(x, torch.tensor(cu, dtype=torch.int32), position_embeddings=(cos, sin))
    save("vision_attention", **out)


def main():
    t0 = time.time()
    ns = load_reference_fla()
    ns = load_reference_std(ns)
    only = set(sys.argv[1:])
    steps = [("gdn", gen_gdn_ops), ("conv", gen_conv_norm), ("swa", gen_swa), ("cache", gen_cache_traces),
             ("layers", gen_layers), ("swa_d128", gen_swa_d128), ("stack_realdims", gen_stack_realdims),
             ("vision", gen_vision)]
    for name, fn in steps:
        if only and name not in only:
            continue
        fn(ns)
    print(f"done in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
