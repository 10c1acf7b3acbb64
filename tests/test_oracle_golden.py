"""CPU: pin the oracle against fixtures produced by executing the reference
(tests/golden/gen_golden.py).  Tolerances: fp32 RMS-relative <= 2e-6 for op-level
float fixtures (the reference kernels ran in fp32 under the Triton interpreter),
bit-exact for integers."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rms_rel
from oracle import cache as ocache
from oracle import gdn, model, swa

GDN_CASES = ["rec_T1_h0", "rec_T7_h0", "rec_T64", "chunk_T65_h0", "chunk_T160", "chunk_T256_h0", "chunk_T100_small"]


@pytest.mark.parametrize("name", GDN_CASES)
def test_gdn_ops_match_reference(name):
    z = load_golden("gdn_" + name)
    h0 = z.get("h0")
    for fn in (gdn.gdn_recurrent, gdn.gdn_chunk):       # both forms are the same function of the inputs
        o, S = fn(z["q"], z["k"], z["v"], z["g"], z["beta"], initial_state=h0)
        assert rms_rel(z["o"], o) < 2e-6, (fn.__name__, rms_rel(z["o"], o))
        assert rms_rel(z["ht"], S) < 2e-6, (fn.__name__, rms_rel(z["ht"], S))


def test_gdn_chained_bf16_state():
    z = load_golden("gdn_chunk_chained_2x128_bf16state")
    sl = lambda a, b: [z[n][:, a:b] for n in ("q", "k", "v", "g", "beta")]  # noqa: E731
    o1, s1 = gdn.gdn_chunk(*sl(0, 128), initial_state=z["h0"])
    s1r = s1.to(torch.bfloat16).float()
    # the reference's mid state differs from ours at fp32 rounding level -> compare after bf16 rounding loosely
    assert rms_rel(z["s_mid"], s1r) < 2e-3
    o2, s2 = gdn.gdn_chunk(*sl(128, 256), initial_state=z["s_mid"])
    assert rms_rel(z["o"], torch.cat([o1, o2], 1)) < 2e-6
    assert rms_rel(z["ht"], s2) < 2e-6


def test_gdn_fp16_rounding_emulation():
    """The reference's reduced-precision path (public wrapper, fp16) vs the oracle's
    rounding-point emulation: same inputs, error well inside fla's 5e-3 convention."""
    z = load_golden("gdn_chunk_T160_fp16")
    args = [torch.from_numpy(np.asarray(z[n])).float() if not torch.is_tensor(z[n]) else z[n].float()
            for n in ("q_f16", "k_f16", "v_f16")]
    beta = z["beta_f16"].float()
    o, S = gdn.gdn_chunk(args[0], args[1], args[2], z["g"], beta, initial_state=z["h0"], rounding=torch.float16)
    assert rms_rel(z["o"], o) < 1.5e-3, rms_rel(z["o"], o)
    assert rms_rel(z["ht"], S) < 1.5e-3, rms_rel(z["ht"], S)
    o32, S32 = gdn.gdn_chunk(args[0], args[1], args[2], z["g"], beta, initial_state=z["h0"])
    assert rms_rel(z["o"], o32) < 5e-3 and rms_rel(z["ht"], S32) < 5e-3


def test_short_conv_matches_reference():
    z = load_golden("short_conv")
    w = z["weight"].reshape(-1, 4)
    y, st = gdn.short_conv(z["x"], w, None)
    assert rms_rel(z["y"], y) < 2e-6 and torch.equal(st, z["state"])
    c = st
    for i in range(5):
        yi, c = gdn.short_conv(z["xs"][i], w, c)
        assert rms_rel(z["ys"][i], yi) < 2e-6
        assert torch.equal(c, z["states"][i])
    y3, st3 = gdn.short_conv(z["x3"], w, None)
    assert rms_rel(z["y3"], y3) < 2e-6 and torch.equal(st3, z["state3"])


def test_short_conv_carry_in_is_split_invariant():
    """Carry-in semantics: conv over one long input == conv over
    its pieces with the state carried, for any split."""
    torch.manual_seed(0)
    x = torch.randn(2, 37, 24)
    w = torch.randn(24, 4)
    y_full, s_full = gdn.short_conv(x, w, None)
    for cuts in ([5, 1, 1, 30], [2, 35], [1] * 37, [20, 17]):
        st, ys, pos = None, [], 0
        for n in cuts:
            yi, st = gdn.short_conv(x[:, pos:pos + n], w, st)
            ys.append(yi)
            pos += n
        assert torch.allclose(torch.cat(ys, 1), y_full, atol=1e-6)
        assert torch.equal(st, s_full)


def test_gdn_varlen_matches_reference():
    """cu_seqlens inputs (fixture: the vendored kernels with offsets / indices, gen_golden_options.py): by definition the
    sequences are independent, so the oracle is the per-sequence rule on each slice with the sequence's own state."""
    z = load_golden("gdn_varlen")
    cu = [int(x) for x in z["cu_seqlens"]]
    for mode in ("chunk", "recurrent"):
        for i, (a, b) in enumerate(zip(cu[:-1], cu[1:])):
            fn = gdn.gdn_chunk if mode == "chunk" else gdn.gdn_recurrent
            o, ht = fn(z["q"][:, a:b], z["k"][:, a:b], z["v"][:, a:b], z["g"][:, a:b], z["beta"][:, a:b], initial_state=z["h0"][i:i + 1])
            assert rms_rel(z["o_" + mode][:, a:b], o) < 2e-5, (mode, i)
            assert rms_rel(z["ht_" + mode][i:i + 1], ht) < 2e-5, (mode, i)


def test_non_default_module_options_match_reference():
    """ShortConvolution(bias=True) and FusedRMSNormGated(elementwise_affine=False) / residual= / prenorm= / residual_in_fp32=
    (fixtures of tests/golden/gen_golden_options.py: the reference's own modules)."""
    z = load_golden("short_conv_bias")
    w, b = z["weight"].reshape(-1, 4), z["bias"]
    y, st = gdn.short_conv(z["x"], w, None, bias=b)
    assert rms_rel(z["y"], y) < 2e-6 and torch.equal(st, z["state"])
    c = st
    for i in range(z["xs"].shape[0]):
        yi, c = gdn.short_conv(z["xs"][i], w, c, bias=b)
        assert rms_rel(z["ys"][i], yi) < 2e-6 and torch.equal(c, z["states"][i])
    z = load_golden("rmsnorm_gate_options")
    eps = float(z["eps"])
    assert rms_rel(z["y_no_affine"], gdn.rmsnorm_swish_gate(z["x"], z["gate"], None, eps)) < 2e-6
    assert rms_rel(z["y_residual"], gdn.rmsnorm_swish_gate(z["x"], z["gate"], z["weight"], eps, residual=z["residual"])) < 2e-6
    y, r = gdn.rmsnorm_swish_gate(z["x"], z["gate"], z["weight"], eps, residual=z["residual"], return_residual=True)
    assert rms_rel(z["y_prenorm"], y) < 2e-6 and torch.equal(r, z["residual_out"])
    y, r = gdn.rmsnorm_swish_gate(z["x"], z["gate"], z["weight"], eps, return_residual=True)
    assert rms_rel(z["y_prenorm_fp32"], y) < 2e-6 and torch.equal(r, z["residual_out_fp32"]) and r.dtype == torch.float32


def test_rmsnorm_gate_matches_reference():
    z = load_golden("rmsnorm_gate")
    y = gdn.rmsnorm_swish_gate(z["x"], z["gate"], z["weight"], float(z["eps"]))
    assert rms_rel(z["y"], y) < 2e-6


def test_swa_band_bit_exact():
    z = load_golden("swa_band")
    off = 0
    for W, seen, T, n_prev in zip(z["W"].tolist(), z["seen"].tolist(), z["T"].tolist(), z["n_prev"].tolist()):
        assert swa.n_prev_keys(W, seen) == n_prev
        lo, hi = swa.window_bounds(n_prev, T, W)
        assert np.array_equal(lo, z["lo"][off:off + T].numpy()), (W, seen, T)
        assert np.array_equal(hi, z["hi"][off:off + T].numpy()), (W, seen, T)
        off += T
    assert off == len(z["lo"]) and off > 300


def test_swa_attention_matches_reference():
    z = load_golden("swa_attention")
    W = int(z["W"])
    for name in ("empty_Tlt", "empty_Tgt", "cached_full", "decode", "cached_part"):
        out = swa.swa_attention(z[name + "_q"], z[name + "_k"], z[name + "_v"], int(z[name + "_n_prev"]), W)
        assert rms_rel(z[name + "_out"], out) < 2e-6, name


def test_swa_attention_d128_matches_reference():
    """The head shape the HIP kernel is built for (d = 128, GQA group 8 / 16-over-2 heads), windows smaller and larger
    than the call."""
    z = load_golden("swa_attention_d128")
    for name in [str(n) for n in z["names"]]:
        out = swa.swa_attention(z[name + "_q"], z[name + "_k"], z[name + "_v"], int(z[name + "_n_prev"]), int(z[name + "_W"]))
        assert rms_rel(z[name + "_out"], out) < 2e-6, name


def test_vision_attention_matches_reference():
    """the vision rotary embedding (bf16 in / bf16 out: BIT-exact against the reference's
    apply_rotary_pos_emb_vision), the rotary tables from (h, w) patch coordinates, the per-window attention at head_dim 80
    (windows / one full segment / ragged with an empty and a one-patch segment) and the whole attention block."""
    from oracle import vision
    z = load_golden("vision_attention")
    cos, sin = vision.vision_rotary_tables(z["pos_hw"], 80)
    assert torch.equal(cos, z["cos"]) and torch.equal(sin, z["sin"])
    q, k, v = (z[n].to(torch.bfloat16) for n in ("q", "k", "v"))
    qr, kr = vision.apply_rotary_pos_emb_vision(q, k, z["cos"], z["sin"])
    assert torch.equal(qr.float(), z["q_rot"]) and torch.equal(kr.float(), z["k_rot"])
    for name in ("windows", "full", "ragged"):
        out = vision.segment_attention(z["q_rot"], z["k_rot"], z["v"], z[name + "_cu"].tolist())
        assert rms_rel(z[name + "_out"], out) < 2e-6, name
    for name in ("windows", "full"):
        out = vision.vision_attention_block(z["mod_x"], z["mod_qkv_weight"], z["mod_qkv_bias"], z["mod_proj_weight"],
                                            z["mod_proj_bias"], z[name + "_cu"].tolist(), z["cos"], z["sin"], 4)
        assert rms_rel(z[f"mod_{name}_out"], out) < 2e-6, name


def test_mrope_matches_reference():
    z = load_golden("mrope")
    cos, sin = swa.rotary_cos_sin(z["position_ids"], 128, float(z["theta"]))
    assert torch.allclose(cos, z["cos"], atol=2e-6) and torch.allclose(sin, z["sin"], atol=2e-6)
    qe, ke = swa.apply_mrope(z["q"], z["k"], cos, sin, z["mrope_section"].tolist())
    assert rms_rel(z["q_out"], qe) < 2e-6 and rms_rel(z["k_out"], ke) < 2e-6


def test_cache_integer_traces_bit_exact():
    z = load_golden("cache_traces")
    keys = sorted(k[:-6] for k in z if k.endswith("_steps"))
    assert len(keys) == 4
    for key in keys:
        steps, W = z[key + "_steps"].tolist(), int(z[key + "_W"])
        ref = z[key + "_trace"].numpy()
        got = np.array([t[:5] for t in ocache.swa_trace(W, steps)], dtype=np.int64)
        assert np.array_equal(ref, got), key
    # survey's probed trace
    tr = ocache.swa_trace(8, [5, 1, 1, 1, 6, 1, 20])
    assert [t[0] for t in tr] == [5, 6, 7, 8, 13, 8, 27]
    assert [t[1] for t in tr] == [5, 6, 7, 7, 7, 7, 7]
    assert [(t[3], t[4]) for t in tr] == [(5, 0), (6, 0), (7, 0), (8, 0), (13, 1), (8, 7), (27, 8)]


def _tiny():
    z = load_golden("tiny_stack")
    lt = [str(x) for x in z["layer_types"]]
    cfg = model.OracleConfig(hidden_size=64, intermediate_size=96, num_attention_heads=4, num_key_value_heads=2,
                             num_linear_heads=4, linear_head_dim=16, expand_v=2, conv_size=4, sliding_window=8,
                             rope_theta=1e4, mrope_section=[2, 3, 3], layer_types=lt)
    params = {k[2:]: v for k, v in z.items() if k.startswith("w.")}
    return z, cfg, params


def _pos(start, T):
    return torch.arange(start, start + T)[None, None, :].expand(3, 1, T).contiguous()


def test_tiny_stack_harness_matches_reference():
    """Row H: prefill 70 -> 2 streaming frames of 20 -> clone -> greedy decode 6 ->
    one more frame on the original; hidden states, tokens, logits and final cache content."""
    z, cfg, p = _tiny()
    embed = p["embed_tokens.weight"]
    cache = model.new_cache(cfg)
    h0 = model.text_stack(p, embed[z["s.ids0"]], _pos(0, 70), cfg, cache)
    assert rms_rel(z["s.h0"], h0) < 1e-5, rms_rel(z["s.h0"], h0)
    pos = 70
    for f in range(2):
        h = model.text_stack(p, z["s.frames"][f], _pos(pos, 20), cfg, cache)
        assert rms_rel(z["s.h_frames"][f], h) < 1e-5, f
        pos += 20
    qa = model.clone_cache(cache)
    tok = int((h[0, -1] @ embed.T).argmax())
    qpos = pos
    for step in range(6):
        hq = model.text_stack(p, embed[torch.tensor([[tok]])], _pos(qpos, 1), cfg, qa)
        lg = hq[0, -1] @ embed.T
        assert rms_rel(z["s.decode_logits"][step], lg) < 1e-5
        tok = int(lg.argmax())
        assert tok == int(z["s.decode_tokens"][step])
        qpos += 1
    h3 = model.text_stack(p, z["s.frames"][2], _pos(pos, 20), cfg, cache)
    assert rms_rel(z["s.h_frame2_after_clone"], h3) < 1e-5
    sw = cache[0]
    assert sw.counters.size == int(z["s.swa_size"]) and sw.counters.cumulative_length == int(z["s.swa_cum"])
    assert rms_rel(z["s.swa_keys"], sw.k) < 1e-5 and rms_rel(z["s.swa_values"], sw.v) < 1e-5
    assert rms_rel(z["s.gdn1_recurrent"], cache[1].recurrent) < 1e-5
    assert rms_rel(z["s.gdn1_conv_q"], cache[1].conv[0]) < 1e-6
    assert cache[1].counters.seq_len == int(z["s.gdn1_seq_len"])


def _realdims():
    z = load_golden("stack_realdims")
    lt = [str(x) for x in z["layer_types"]]
    cfg = model.OracleConfig(hidden_size=256, intermediate_size=128, num_attention_heads=2, num_key_value_heads=1,
                             num_linear_heads=2, linear_head_dim=128, expand_v=2, conv_size=4, sliding_window=96,
                             rope_theta=1e6, mrope_section=[16, 24, 24], layer_types=lt)
    params = {k[2:]: v for k, v in z.items() if k.startswith("w.")}
    return z, cfg, params


def test_stack_realdims_matches_reference():
    """The same harness scenario as tiny_stack on reference decoder layers with the REAL head shapes (head_dim 128,
    K = 128, V = 256, mrope [16,24,24], window 96 that wraps during the stream)."""
    z, cfg, p = _realdims()
    embed = p["embed_tokens.weight"]
    cache = model.new_cache(cfg)
    h0 = model.text_stack(p, embed[z["s.ids0"]], _pos(0, 70), cfg, cache)
    assert rms_rel(z["s.h0"], h0) < 1e-5, rms_rel(z["s.h0"], h0)
    pos = 70
    for f in range(2):
        h = model.text_stack(p, z["s.frames"][f], _pos(pos, 20), cfg, cache)
        assert rms_rel(z["s.h_frames"][f], h) < 1e-5, f
        pos += 20
    qa = model.clone_cache(cache)
    tok = int((h[0, -1] @ embed.T).argmax())
    assert tok == int(z["s.first_token"])
    qpos = pos
    for step in range(6):
        hq = model.text_stack(p, embed[torch.tensor([[tok]])], _pos(qpos, 1), cfg, qa)
        lg = hq[0, -1] @ embed.T
        assert rms_rel(z["s.decode_logits"][step], lg) < 1e-5
        tok = int(lg.argmax())
        assert tok == int(z["s.decode_tokens"][step])
        qpos += 1
    h3 = model.text_stack(p, z["s.frames"][2], _pos(pos, 20), cfg, cache)
    assert rms_rel(z["s.h_frame2_after_clone"], h3) < 1e-5
    sw = cache[0]
    assert sw.counters.size == int(z["s.swa_size"]) == 95 and sw.counters.cumulative_length == int(z["s.swa_cum"]) == 130
    assert rms_rel(z["s.swa_keys"], sw.k) < 1e-5 and rms_rel(z["s.swa_values"], sw.v) < 1e-5
    assert rms_rel(z["s.gdn1_recurrent"], cache[1].recurrent) < 1e-5
    assert rms_rel(z["s.gdn1_conv_q"], cache[1].conv[0]) < 1e-6
    # single mixers at the module boundary
    o = model.gdn_layer(model._sub(p, "layers.1.self_attn."), z["m.x_gdn"], cfg, None)
    assert rms_rel(z["m.o_gdn"], o) < 1e-5
    cos, sin = swa.rotary_cos_sin(z["m.pos3"], cfg.head_dim, cfg.rope_theta)
    o = model.swa_layer(model._sub(p, "layers.0.self_attn."), z["m.x_swa"], cos, sin, cfg, None)
    assert rms_rel(z["m.o_swa"], o) < 1e-5


def test_tiny_mixers_match_reference():
    z, cfg, p = _tiny()
    zm = load_golden("tiny_mixers")
    pg = model._sub(p, "layers.1.self_attn.")
    o = model.gdn_layer(pg, zm["x_gdn"], cfg, None)
    assert rms_rel(zm["o_gdn"], o) < 1e-5, rms_rel(zm["o_gdn"], o)
    ps = model._sub(p, "layers.0.self_attn.")
    cos, sin = swa.rotary_cos_sin(zm["pos3"], cfg.head_dim, cfg.rope_theta)
    o = model.swa_layer(ps, zm["x_swa"], cos, sin, cfg, None)
    assert rms_rel(zm["o_swa"], o) < 1e-5, rms_rel(zm["o_swa"], o)


def test_configs0_plumbing_run_on_the_cpu_path():
    """BASELINE.json configs[0]: "mllm_demo.json single-image greedy decode on CPU eager path
    (plumbing, no GPU)".  The workload of sample 0 after the processor -- 64 random ids + 256 image-placeholder embeds
    with (t, h, w) M-RoPE positions -- through the CPU restatement of the path with the infinitevl_standard layer
    semantics (one 4-layer period: 1 sliding-window + 3 Gated DeltaNet decoder layers at InfiniteVL-3B's head shapes,
    window 4096), greedy 16 tokens.  Asserts plumbing only: shapes, finiteness, counters, determinism.  The GPU twin
    (tests/test_gpu_parity.py::test_configs0_plumbing_hip_stack_vs_cpu_path) runs the HIP stack on the same workload."""
    import parity
    lt = ["sliding_attention", "linear_attention", "linear_attention", "linear_attention"]
    oc = model.OracleConfig(hidden_size=2048, intermediate_size=2048, num_attention_heads=16, num_key_value_heads=2,
                            num_linear_heads=16, linear_head_dim=128, expand_v=2.0, conv_size=4, sliding_window=4096,
                            rope_theta=1e6, mrope_section=[16, 24, 24], layer_types=lt)
    params = model.random_params(oc, seed=0, vocab=4096)
    toks, logits, cache = parity.configs0_oracle_run(oc, params)
    assert len(toks) == 16 and all(0 <= t < 4096 for t in toks)
    assert tuple(logits.shape) == (16, 4096) and bool(torch.isfinite(logits).all())
    assert cache[0].counters.cumulative_length == 320 + 16 and cache[0].counters.size == 320 + 16    # window 4096 not full
    assert cache[1].counters.seq_len > 0
    toks2, _, _ = parity.configs0_oracle_run(oc, params)
    assert toks == toks2
