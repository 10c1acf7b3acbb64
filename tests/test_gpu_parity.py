"""GPU (-m gpu): the HIP path, called through the C ABI, against (a) the committed golden fixtures that
were produced by executing the reference, (b) the CPU oracle on the same seeded inputs, (c) size-
independent properties at BASELINE.json's full sizes.

Stated tolerances (SURVEY.md section 4 / fla:ops/utils/testing.py: RMS-relative error):
  * bf16-I/O kernels vs an exact fp32 reference result:            <= 5e-3  (fla's fwd convention)
  * same vs the oracle run with the reference's bf16 rounding points: <= 5e-4 (outputs), 1e-5..2e-3 (state,
    2e-3 only when the state itself is stored in bf16)
  * integers (window band, counters, ring placement) and M-RoPE: bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rms_rel
from oracle import gdn as ogdn
from oracle import swa as oswa
import parity


def _free_port():
    """A TCP port that is free right now on 127.0.0.1 (a pid-derived constant collided now and then with a socket of an
    earlier run still in TIME_WAIT)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16)


@pytest.fixture(scope="module", autouse=True)
def _lib():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import infinitevl_amd
    infinitevl_amd.load_library()          # fails loudly when the HIP extension is missing
    yield


# ---------------------------------------------------------------------------------------------
# Gated DeltaNet operator
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["rec_T1_h0", "rec_T7_h0", "rec_T64", "chunk_T65_h0", "chunk_T160", "chunk_T256_h0"])
def test_gdn_golden_reference_vectors(name):
    from infinitevl_amd import ops
    z = load_golden("gdn_" + name)
    fn = ops.fused_recurrent_gated_delta_rule if name.startswith("rec") else ops.chunk_gated_delta_rule
    h0 = z.get("h0")
    o, ht = fn(bf(z["q"]).to(DEV), bf(z["k"]).to(DEV), bf(z["v"]).to(DEV), z["g"].to(DEV), bf(z["beta"]).to(DEV),
               initial_state=None if h0 is None else h0.to(DEV), output_final_state=True, use_qk_l2norm_in_kernel=True)
    assert ht.dtype == torch.float32 and o.dtype == torch.bfloat16
    assert rms_rel(z["o"], o.float().cpu()) < 5e-3
    assert rms_rel(z["ht"], ht.cpu()) < 5e-3


def test_gdn_golden_chained_bf16_state():
    """Q5: state rounded to the cache dtype between calls (kernel writes bf16 in place)."""
    from infinitevl_amd import ops
    z = load_golden("gdn_chunk_chained_2x128_bf16state")
    state = bf(z["h0"]).to(DEV)
    outs = []
    for a, b in ((0, 128), (128, 256)):
        o, _ = ops.chunk_gated_delta_rule(bf(z["q"][:, a:b]).to(DEV), bf(z["k"][:, a:b]).to(DEV), bf(z["v"][:, a:b]).to(DEV),
                                          z["g"][:, a:b].to(DEV), bf(z["beta"][:, a:b]).to(DEV), initial_state=state,
                                          use_qk_l2norm_in_kernel=True, final_state_out=state)
        outs.append(o.float().cpu())
        if a == 0:
            assert rms_rel(z["s_mid"], state.float().cpu()) < 5e-3
    assert rms_rel(z["o"], torch.cat(outs, 1)) < 5e-3
    assert rms_rel(z["ht"], state.float().cpu()) < 6e-3


@pytest.mark.parametrize("mode,B,T,H,h0,sd,inplace", [
    ("recurrent", 1, 1, 2, True, torch.float32, False), ("recurrent", 2, 7, 3, True, torch.float32, False),
    ("recurrent", 1, 64, 2, False, torch.float32, False), ("recurrent", 1, 33, 16, True, torch.bfloat16, True),
    ("recurrent", 1, 70, 2, True, torch.float32, False),
    ("chunk", 1, 64, 1, False, torch.float32, False), ("chunk", 1, 65, 2, True, torch.float32, False),
    ("chunk", 1, 1, 2, True, torch.float32, False), ("chunk", 1, 30, 2, True, torch.float32, False),
    ("chunk", 2, 256, 2, True, torch.float32, False), ("chunk", 1, 257, 16, True, torch.bfloat16, True),
    ("chunk", 1, 1000, 2, False, torch.float32, False),
])
def test_gdn_vs_oracle(mode, B, T, H, h0, sd, inplace):
    r = parity.gdn_op_parity(DEV, mode, B, T, H, seed=T + H, with_h0=h0, state_dtype=sd, inplace_state=inplace)
    assert r["finite"] == 1.0
    assert r["o_vs_exact"] < 5e-3 and r["s_vs_exact"] < 5e-3, r
    assert r["o_vs_bf16model"] < 5e-4, r
    assert r["s_vs_bf16model"] < (2.5e-3 if sd == torch.bfloat16 else 1e-4), r


# fp8 (e4m3) operand variant of the chunk rule -- BASELINE.json configs[4].  The reference has no fp8 path, so the
# tolerance is the build's own statement: e4m3 carries 3 mantissa bits (relative rounding error <= 2^-4 per operand),
# which puts the oracle WITH the same fp8 rounding points at 3.9e-2 RMS-relative from the exact fp32 result on this input
# distribution.  The HIP path must (a) reproduce that fp8-rounding model within 1e-3 (observed 6e-5 .. 9e-5: what remains
# is e4m3 rounding flips caused by fp32 summation order) and (b) stay within 8e-2 of the exact result -- 2x the model's own
# distance.
@pytest.mark.parametrize("B,T,H,h0,sd,inplace", [
    (1, 65, 2, True, torch.float32, False), (1, 256, 16, True, torch.bfloat16, True), (2, 300, 3, False, torch.float32, False),
    (1, 1000, 2, True, torch.float32, False),
])
def test_gdn_fp8_vs_oracle(B, T, H, h0, sd, inplace):
    r = parity.gdn_op_parity(DEV, "chunk", B, T, H, seed=T + H, with_h0=h0, state_dtype=sd, inplace_state=inplace,
                             mma_dtype="fp8_e4m3")
    assert r["finite"] == 1.0
    assert r["o_vs_fp8model"] < 1e-3 and r["s_vs_fp8model"] < (3e-3 if sd == torch.bfloat16 else 1e-3), r   # bf16-stored state: 2^-9
    assert r["o_vs_exact"] < 8e-2 and r["s_vs_exact"] < 8e-2, r
    assert r["o_vs_exact"] > 5e-3, ("the fp8 variant should not be as accurate as the bf16 one", r)


@pytest.mark.parametrize("seed", range(int(os.environ.get("IVL_GDN_FUZZ", "24"))))
def test_gdn_random_shapes_vs_oracle(seed):
    """Randomised differential test of the chunk rule (single-launch step form, persistent long-call form, segment seams,
    ragged last chunks, batches, bf16 / fp32 carried state, in-place state, bf16 and fp8 operands) against the oracle with the
    reference's rounding points.  IVL_GDN_FUZZ=N runs N seeds (default 24)."""
    import random
    r = random.Random(seed)
    H = r.choice([1, 2, 3, 16])
    B = r.choice([1, 1, 2, 3])
    T = r.choice([r.randint(1, 64), r.randint(65, 300), r.randint(257, 1500)])
    if H == 16 or B == 3:
        T = min(T, 420)                                                     # keeps the CPU oracle at seconds
    sd = r.choice([torch.float32, torch.bfloat16])
    h0, inplace, fp8 = r.random() < 0.7, r.random() < 0.5, r.random() < 0.3
    res = parity.gdn_op_parity(DEV, "chunk", B, T, H, seed=7000 + seed, with_h0=h0, state_dtype=sd, inplace_state=inplace,
                               mma_dtype="fp8_e4m3" if fp8 else None)
    case = dict(B=B, T=T, H=H, h0=h0, sd=str(sd), inplace=inplace, fp8=fp8)
    assert res["finite"] == 1.0, (res, case)
    # Bounds of the fixed cases, widened where 750 seeds showed what a random input can do WITHOUT a defect (seeds 7, 97 and
    # 446 taken apart on the GPU per head, row and chunk): (a) a single operand element on a rounding tie resolved the other way by the kernel's fp32 summation
    # order moves one state row (bf16: seeds 7, 97: 2.4e-4 / 4.9e-4 of the state) or one chunk of one head (e4m3, 2^-4 per
    # flip: seed 446: 1.1e-2 inside chunk 1 of head 10, nothing before, nothing after, state untouched = 1.4e-3 of the
    # output; bf16 at T = 19, where one flipped row weighs more: seed 189, 6.7e-4, state 6.6e-5); (b) at T = 3 the reference-rounding model itself is 6.2e-3 from exact fp32
    # (seed 367: the kernel equals that model bit for bit) -- so the distance to exact is bounded against the model's own.
    if fp8:
        assert res["o_vs_fp8model"] < 3e-3 and res["s_vs_fp8model"] < 3e-3, (res, case)
        assert res["o_vs_exact"] < 8e-2 and res["s_vs_exact"] < 8e-2, (res, case)
    else:
        assert res["o_vs_exact"] < max(5e-3, 1.1 * res["o_bf16model_vs_exact"] + 2e-4), (res, case)
        assert res["s_vs_exact"] < max(5e-3, 1.1 * res["s_bf16model_vs_exact"] + 2e-4), (res, case)
        assert res["o_vs_bf16model"] < 1e-3, (res, case)
        assert res["s_vs_bf16model"] < (2.5e-3 if sd == torch.bfloat16 else 1e-3), (res, case)


def test_gdn_long_call_uses_segments_and_matches_chained_calls():
    """Full-size property (H=16, T=8192+100 > one 4096-token workspace segment): one long call ==
    the same tokens fed as 256-token calls with the fp32 state carried (split invariance)."""
    from infinitevl_amd import ops
    B, T, H = 1, 8292, 16
    q, k, v, g, beta, h0 = parity.gdn_inputs(5, B, T, H)
    qd, kd, vd, bd, gd = bf(q).to(DEV), bf(k).to(DEV), bf(v).to(DEV), bf(beta).to(DEV), g.to(DEV)
    o1, s1 = ops.chunk_gated_delta_rule(qd, kd, vd, gd, bd, initial_state=h0.to(DEV), output_final_state=True,
                                        use_qk_l2norm_in_kernel=True)
    st = h0.to(DEV)
    outs = []
    for a in range(0, T, 256):
        b_ = min(T, a + 256)
        o, st = ops.chunk_gated_delta_rule(qd[:, a:b_], kd[:, a:b_], vd[:, a:b_], gd[:, a:b_], bd[:, a:b_],
                                           initial_state=st, output_final_state=True, use_qk_l2norm_in_kernel=True)
        outs.append(o)
    o2 = torch.cat(outs, 1)
    assert torch.isfinite(o1.float()).all()
    # chunk boundaries coincide (256 = 4 chunks) -> identical arithmetic
    assert torch.equal(o1, o2) and torch.equal(s1, st)


def test_gdn_chunk_equals_recurrent_kernel_at_full_width():
    """Two independent kernels, same function (H=16 heads, T=64: the mode-switch boundary, Q7)."""
    from infinitevl_amd import ops
    q, k, v, g, beta, h0 = parity.gdn_inputs(9, 2, 64, 16)
    args = (bf(q).to(DEV), bf(k).to(DEV), bf(v).to(DEV), g.to(DEV), bf(beta).to(DEV))
    o1, s1 = ops.chunk_gated_delta_rule(*args, initial_state=h0.to(DEV), output_final_state=True, use_qk_l2norm_in_kernel=True)
    o2, s2 = ops.fused_recurrent_gated_delta_rule(*args, initial_state=h0.to(DEV), output_final_state=True,
                                                  use_qk_l2norm_in_kernel=True)
    assert rms_rel(o2.float().cpu(), o1.float().cpu()) < 5e-3 and rms_rel(s2.cpu(), s1.cpu()) < 5e-3


def test_gdn_fp16_activations_reference_vectors():
    """IEEE-half activations (fla accepts them: chunk.py:352 refuses fp32 only): both operators on fp16 tensors against the
    reference's own fp16 chunk result (fixture gdn_chunk_T160_fp16, produced through fla's public wrapper) -- fp16 in, fp16 out,
    fp32 state; <= 5e-3 like every bf16-I/O kernel (the package evaluates the rule token by token in fp32 on the fp16 inputs: no
    intermediate roundings, so it sits at the fp16 noise floor of the reference's chunk form) -- and against the oracle's exact
    recurrent rule on the same fp16-valued inputs."""
    from infinitevl_amd import ops
    z = load_golden("gdn_chunk_T160_fp16")
    q, k, v, beta = (z[n].to(DEV, torch.float16) for n in ("q_f16", "k_f16", "v_f16", "beta_f16"))
    g, h0 = z["g"].to(DEV), z["h0"].to(DEV)
    o_ex, s_ex = ogdn.gdn_recurrent(z["q_f16"].float(), z["k_f16"].float(), z["v_f16"].float(), z["g"], z["beta_f16"].float(), initial_state=z["h0"],
                                    qk_round_dtype=torch.float16)
    for fn in (ops.chunk_gated_delta_rule, ops.fused_recurrent_gated_delta_rule):
        o, ht = fn(q, k, v, g, beta, initial_state=h0, output_final_state=True, use_qk_l2norm_in_kernel=True)
        assert o.dtype == torch.float16 and ht.dtype == torch.float32
        assert rms_rel(z["o"], o.float().cpu()) < 5e-3 and rms_rel(z["ht"], ht.cpu()) < 5e-3, fn.__name__
        assert rms_rel(o_ex, o.float().cpu()) < 1e-3 and rms_rel(s_ex, ht.cpu()) < 1e-4, fn.__name__
    with pytest.raises(ValueError):
        ops.chunk_gated_delta_rule(q, k, v, g, beta, mma_dtype="fp8_e4m3")
    with pytest.raises(ValueError):
        ops.chunk_gated_delta_rule(q.float(), k.float(), v.float(), g, beta.float())          # fp32 stays refused (chunk.py:352)


def test_gdn_varlen_reference_vectors():
    """cu_seqlens inputs of both GDN operators against the reference's own outputs (4 flattened sequences of 70 / 100 / 1 / 79
    tokens, one initial state each) and, bit for bit, against the same sequences as separate calls; fla's argument checks."""
    from infinitevl_amd import ops
    z = load_golden("gdn_varlen")
    cu = torch.from_numpy(np.asarray(z["cu_seqlens"])).to(DEV)
    q, k, v, beta = (bf(z[n]).to(DEV) for n in ("q", "k", "v", "beta"))
    g, h0 = z["g"].to(DEV), z["h0"].to(DEV)
    for mode, fn in (("chunk", ops.chunk_gated_delta_rule), ("recurrent", ops.fused_recurrent_gated_delta_rule)):
        o, ht = fn(q, k, v, g, beta, initial_state=h0, output_final_state=True, cu_seqlens=cu, use_qk_l2norm_in_kernel=True)
        assert tuple(ht.shape) == (4, 2, 128, 256) and ht.dtype == torch.float32
        assert rms_rel(z["o_" + mode], o.float().cpu()) < 5e-3 and rms_rel(z["ht_" + mode], ht.cpu()) < 5e-3, mode
        bounds = [int(x) for x in z["cu_seqlens"]]
        for i, (a, b_) in enumerate(zip(bounds[:-1], bounds[1:])):
            oi, hti = fn(q[:, a:b_].contiguous(), k[:, a:b_].contiguous(), v[:, a:b_].contiguous(), g[:, a:b_].contiguous(),
                         beta[:, a:b_].contiguous(), initial_state=h0[i:i + 1], output_final_state=True, use_qk_l2norm_in_kernel=True)
            assert torch.equal(oi, o[:, a:b_]) and torch.equal(hti, ht[i:i + 1]), (mode, i)
    with pytest.raises(ValueError, match="batch size is expected to be 1"):
        ops.chunk_gated_delta_rule(q.repeat(2, 1, 1, 1), k.repeat(2, 1, 1, 1), v.repeat(2, 1, 1, 1), g.repeat(2, 1, 1), beta.repeat(2, 1, 1), cu_seqlens=cu)
    with pytest.raises(ValueError, match="number of initial states"):
        ops.chunk_gated_delta_rule(q, k, v, g, beta, initial_state=h0[:3], cu_seqlens=cu)
    with pytest.raises(ValueError, match="cu_seqlens must rise"):
        ops.chunk_gated_delta_rule(q, k, v, g, beta, cu_seqlens=torch.tensor([0, 70, 200], device=DEV))
    # an empty sequence passes its state through; the short convolution takes the same offsets
    cu2 = torch.tensor([0, 70, 70, 250], device=DEV)
    o_c, _ = ops.chunk_gated_delta_rule(q, k, v, g, beta, initial_state=h0, output_final_state=True, cu_seqlens=cu, use_qk_l2norm_in_kernel=True)
    o2, ht2 = ops.chunk_gated_delta_rule(q, k, v, g, beta, initial_state=h0[:3], output_final_state=True, cu_seqlens=cu2, use_qk_l2norm_in_kernel=True)
    assert torch.equal(ht2[1], h0[1]) and torch.equal(o2[:, :70], o_c[:, :70])
    conv = ops.ShortConvolution(64, 4, activation="silu").to(DEV, torch.bfloat16)
    x = bf(torch.randn(1, 250, 64, device=DEV))
    y, st = conv(x, output_final_state=True, cu_seqlens=cu)
    for i, (a, b_) in enumerate(zip(bounds[:-1], bounds[1:])):
        yi, sti = conv(x[:, a:b_].contiguous(), output_final_state=True)
        assert torch.equal(yi, y[:, a:b_]) and torch.equal(sti[0], st[i])


def test_gdn_head_first_layout_equals_time_major():
    """fla's deprecated head_first=True layout ([B,H,T,.], chunk.py:361-373) is accepted and rearranged."""
    from infinitevl_amd import ops
    q, k, v, g, beta, h0 = parity.gdn_inputs(5, 2, 70, 3)
    qd, kd, vd, bd = (x.to(DEV, torch.bfloat16) for x in (q, k, v, beta))
    gd = g.to(DEV)
    for fn in (ops.chunk_gated_delta_rule, ops.fused_recurrent_gated_delta_rule):
        o_tm, s_tm = fn(qd, kd, vd, gd, bd, output_final_state=True, use_qk_l2norm_in_kernel=True)
        o_hf, s_hf = fn(qd.transpose(1, 2), kd.transpose(1, 2), vd.transpose(1, 2), gd.transpose(1, 2),
                        bd.transpose(1, 2), None, None, True, None, True, True)       # positional, fla order
        assert o_hf.shape == (2, 3, 70, 256) and torch.equal(o_hf.transpose(1, 2), o_tm) and torch.equal(s_hf, s_tm)


def test_gdn_head_count_mismatch_raises():
    from infinitevl_amd import ops
    q = bf(torch.randn(1, 8, 4, 128)).to(DEV)
    k = bf(torch.randn(1, 8, 2, 128)).to(DEV)
    v = bf(torch.randn(1, 8, 2, 256)).to(DEV)
    g = torch.zeros(1, 8, 2, device=DEV)
    with pytest.raises(ValueError):
        ops.chunk_gated_delta_rule(q, k, v, g, bf(g), use_qk_l2norm_in_kernel=True)


def test_gdn_error_behaviour():
    from infinitevl_amd import ops
    z = torch.zeros(2, 4, 2, 128, dtype=torch.bfloat16, device=DEV)
    v = torch.zeros(2, 4, 2, 256, dtype=torch.bfloat16, device=DEV)
    g = torch.zeros(2, 4, 2, device=DEV)
    b = torch.zeros(2, 4, 2, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ValueError, match="batch size is expected to be 1"):          # chunk.py:355-360
        ops.chunk_gated_delta_rule(z, z, v, g, b, cu_seqlens=torch.tensor([0, 4, 8], device=DEV))
    with pytest.raises(AssertionError):                                               # chunk.py:353
        ops.chunk_gated_delta_rule(z, z, v, g, b[..., None])
    with pytest.raises(ValueError):
        ops.chunk_gated_delta_rule(z.float(), z.float(), v.float(), g, b)             # chunk.py:352 (fp32 refused)
    with pytest.raises(ValueError):
        ops.chunk_gated_delta_rule(z[..., :64], z[..., :64], v, g, b)                 # unsupported head shape


# ---------------------------------------------------------------------------------------------
# short conv / gated norm / gate math / rope
# ---------------------------------------------------------------------------------------------
def test_short_conv_golden_and_state_bit_exact():
    from infinitevl_amd import ops
    z = load_golden("short_conv")
    D = z["weight"].shape[0]
    conv = ops.ShortConvolution(D, 4).to(DEV, torch.bfloat16)
    with torch.no_grad():
        conv.weight.copy_(bf(z["weight"]))
    y, cache = conv(bf(z["x"]).to(DEV), cache=None, output_final_state=True)
    assert rms_rel(z["y"], y.float().cpu()) < 5e-3
    assert torch.equal(bf(z["state"]), cache.cpu())                  # state = raw inputs: bit-exact
    for i in range(5):
        yi, cache2 = conv(bf(z["xs"][i]).to(DEV), cache=cache, output_final_state=True)
        assert cache2.data_ptr() == cache.data_ptr()                 # updated in place (convolution.py:213)
        assert rms_rel(z["ys"][i], yi.float().cpu()) < 5e-3
        assert torch.equal(bf(z["states"][i]), cache.cpu())
    y3, st3 = conv(bf(z["x3"]).to(DEV), cache=None, output_final_state=True)
    assert rms_rel(z["y3"], y3.float().cpu()) < 5e-3 and torch.equal(bf(z["state3"]), st3.cpu())


def test_short_conv_carry_in_split_invariance_full_width():
    """D=8192 (q+k+v channels of InfiniteVL-3B), T=1024: one call == any split with the state carried,
    bit for bit (the semantics streaming needs)."""
    from infinitevl_amd import ops
    torch.manual_seed(0)
    D, T = 8192, 1024
    conv = ops.ShortConvolution(D, 4).to(DEV, torch.bfloat16)
    x = bf(torch.randn(2, T, D)).to(DEV)
    y_full, s_full = conv(x, cache=None, output_final_state=True)
    for cuts in ([256] * 4, [1, 2, 3, 1018], [1000, 24], [7] * 146 + [2]):
        st, ys, pos = None, [], 0
        for n in cuts:
            yi, st = conv(x[:, pos:pos + n], cache=st, output_final_state=True)
            ys.append(yi)
            pos += n
        assert torch.equal(torch.cat(ys, 1), y_full) and torch.equal(st, s_full), cuts[:3]
    y_ref, _ = ogdn.short_conv(x[:1, :64].float().cpu(), conv.weight.float().cpu().reshape(D, 4), None)
    assert rms_rel(y_ref, y_full[:1, :64].float().cpu()) < 5e-3


def test_rmsnorm_gate_golden():
    from infinitevl_amd import ops
    z = load_golden("rmsnorm_gate")
    n = ops.FusedRMSNormGated(256, eps=float(z["eps"])).to(DEV, torch.bfloat16)
    with torch.no_grad():
        n.weight.copy_(bf(z["weight"]))
    y = n(bf(z["x"]).to(DEV), bf(z["gate"]).to(DEV))
    assert rms_rel(z["y"], y.float().cpu()) < 5e-3


def test_non_default_module_options_reference_vectors():
    """The options of the fla modules that InfiniteVL leaves at their defaults, against the reference's own outputs
    (tests/golden/gen_golden_options.py): ShortConvolution(bias=True) prefill + steps with the state bit-exact;
    FusedRMSNormGated(elementwise_affine=False), residual=, prenorm= (residual_out returned), residual_in_fp32=."""
    from infinitevl_amd import ops
    z = load_golden("short_conv_bias")
    conv = ops.ShortConvolution(64, 4, bias=True, activation="silu").to(DEV, torch.bfloat16)
    with torch.no_grad():
        conv.weight.copy_(z["weight"].reshape(64, 1, 4))
        conv.bias.copy_(z["bias"])
        y, st = conv(bf(z["x"]).to(DEV), cache=None, output_final_state=True)
        assert rms_rel(z["y"], y.cpu()) < 5e-3 and torch.equal(st.float().cpu(), z["state"])
        for i in range(z["xs"].shape[0]):
            yi, st = conv(bf(z["xs"][i]).to(DEV), cache=st, output_final_state=True)
            assert rms_rel(z["ys"][i], yi.cpu()) < 5e-3 and torch.equal(st.float().cpu(), z["states"][i])
    z = load_golden("rmsnorm_gate_options")
    x, g = bf(z["x"]).to(DEV), bf(z["gate"]).to(DEV)
    plain = ops.FusedRMSNormGated(256, elementwise_affine=False, eps=float(z["eps"])).to(DEV)
    assert plain.weight is None and rms_rel(z["y_no_affine"], plain(x, g).cpu()) < 5e-3
    aff = ops.FusedRMSNormGated(256, eps=float(z["eps"])).to(DEV, torch.bfloat16)
    with torch.no_grad():
        aff.weight.copy_(z["weight"])
        res = z["residual"].to(DEV)                                  # fp32, as the reference was given it
        assert rms_rel(z["y_residual"], aff(x, g, residual=res).cpu()) < 5e-3
        y, r = aff(x, g, residual=res, prenorm=True)
        assert rms_rel(z["y_prenorm"], y.cpu()) < 5e-3 and r.dtype == torch.float32 and torch.equal(r.cpu(), z["residual_out"])
        y, r = aff(x, g, residual=bf(res), prenorm=True)              # bf16 residual stream: residual_out in bf16, the row unrounded
        assert r.dtype == torch.bfloat16 and rms_rel(z["y_prenorm"], y.cpu()) < 5e-3 and rms_rel(z["residual_out"], r.cpu()) < 5e-3
        y, r = aff(x, g, prenorm=True, residual_in_fp32=True)
        assert rms_rel(z["y_prenorm_fp32"], y.cpu()) < 5e-3 and r.dtype == torch.float32 and torch.equal(r.cpu(), z["residual_out_fp32"])
        y, r = aff(x, g, prenorm=True)                                # no residual, same dtype: the input itself comes back (fused_norm_gate.py:155)
        assert r is x or torch.equal(r, x)


def test_gate_math_vs_oracle():
    from infinitevl_amd import ops
    torch.manual_seed(2)
    a, b = bf(torch.randn(2, 50, 16) * 4), bf(torch.randn(2, 50, 16) * 4)
    a[0, 0, 0], a[0, 0, 1] = 30.0, -30.0                      # softplus threshold / underflow branches
    A_log = torch.log(torch.empty(16).uniform_(0.5, 16))
    dt = torch.randn(16) * 0.5
    g, beta = ops.gdn_gate(a.to(DEV), b.to(DEV), A_log.to(DEV), dt.to(DEV))
    g_ref, b_ref = ogdn.gate_math(a, b, A_log, dt)
    assert g.dtype == torch.float32 and rms_rel(g_ref, g.cpu()) < 1e-6
    assert torch.equal(b_ref, beta.cpu())                    # bf16 sigmoid: bit-exact


def test_mrope_golden_and_bit_exact_vs_reference_arithmetic():
    from infinitevl_amd import ops
    z = load_golden("mrope")
    # the fixture holds the reference's fp32 run; its bf16 arithmetic is reproduced bit for bit:
    q, k = bf(z["q"]), bf(z["k"])                           # [B,H,T,d]
    cos, sin = bf(z["cos"]), bf(z["sin"])
    qe, ke = oswa.apply_mrope(q, k, cos, sin, z["mrope_section"].tolist())     # bf16 eager == std:982-983
    qd, kd = q.transpose(1, 2).contiguous().to(DEV), k.transpose(1, 2).contiguous().to(DEV)
    ops.apply_mrope_inplace(qd, kd, cos.to(DEV), sin.to(DEV), z["mrope_section"].tolist())
    assert torch.equal(qe.transpose(1, 2).contiguous(), qd.cpu()) and torch.equal(ke.transpose(1, 2).contiguous(), kd.cpu())
    assert rms_rel(z["q_out"].transpose(1, 2), qd.float().cpu()) < 5e-3


# ---------------------------------------------------------------------------------------------
# sliding-window attention
# ---------------------------------------------------------------------------------------------
SWA_CASES = [
    # B, T, Hq, Hkv, W, seen, via
    (1, 5, 2, 1, 8, 0, "cat"), (2, 19, 4, 2, 8, 0, "cat"), (1, 6, 16, 2, 8, 9, "cat"), (1, 1, 16, 2, 8, 7, "cat"),
    (1, 130, 16, 2, 4096, 0, "cat"), (1, 200, 2, 1, 96, 0, "cat"),
    (1, 70, 2, 1, 96, 250, "ring"), (1, 256, 16, 2, 4096, 4500, "ring"), (1, 256, 16, 2, 4096, 1000, "ring"),
    (1, 1, 16, 2, 4096, 5000, "ring"), (2, 1, 16, 2, 96, 40, "ring"), (1, 3, 16, 2, 96, 500, "ring"),
    (1, 300, 2, 1, 96, 77, "ring"), (1, 64, 16, 2, 8192, 8191, "ring"), (1, 1, 16, 2, 2, 5, "ring"),
    # packed multi-token decode rows over a full window: 64-way split + wide combine
    (1, 4, 16, 2, 4096, 6000, "ring"), (2, 2, 16, 2, 4096, 4500, "ring"), (1, 8, 16, 2, 2048, 1500, "ring"),
]


@pytest.mark.parametrize("B,T,Hq,Hkv,W,seen,via", SWA_CASES)
def test_swa_vs_oracle(B, T, Hq, Hkv, W, seen, via):
    r = parity.swa_op_parity(DEV, B, T, Hq, Hkv, W, seen, seed=T + seen, via=via)
    assert r["finite"] == 1.0 and r["o"] < 5e-3, r


# fp8 (e4m3) decode step -- BASELINE.json configs[4]; the build's own tolerance (the reference has no fp8 path): with q, K,
# V rounded to e4m3 (3 mantissa bits) the oracle sits ~3e-2 RMS-relative from the exact result; the kernel additionally
# rounds the probabilities to e4m3 (relative 2^-4 each, averaged over >= 64 keys per tile), so it must stay within 3e-2 of
# that operand-rounding model (bound 4e-2, observed 2e-2) and within 8e-2 of the exact fp32 result (observed 5e-2, the
# operand-rounding model itself 4.6e-2).  Calls longer than the packed decode tile run bf16.
@pytest.mark.parametrize("B,T,Hq,Hkv,W,seen", [(1, 1, 16, 2, 4096, 5000), (2, 1, 16, 2, 96, 40), (1, 4, 16, 2, 4096, 6000),
                                                (1, 1, 16, 2, 4096, 100), (1, 8, 16, 2, 2048, 1500), (1, 1, 16, 2, 2, 5)])
def test_swa_fp8_decode_vs_oracle(B, T, Hq, Hkv, W, seen):
    r = parity.swa_op_parity(DEV, B, T, Hq, Hkv, W, seen, seed=T + seen, via="ring", mma_dtype="fp8_e4m3")
    assert r["finite"] == 1.0
    assert r["o_vs_fp8model"] < 4e-2 and r["o"] < 8e-2, r
    if W > 2:
        assert r["o"] > 5e-3, ("the fp8 variant should not be as accurate as the bf16 one", r)


def test_swa_fp8_prefill_calls_stay_bf16():
    r = parity.swa_op_parity(DEV, 1, 70, 2, 1, 96, 250, seed=1, via="ring", mma_dtype="fp8_e4m3")
    assert r["o"] < 5e-3, r


@pytest.mark.parametrize("B,T,seen,W", [(1, 256, 4500, 4096), (2, 70, 40, 96), (1, 1, 300, 96), (1, 300, 0, 4096), (1, 130, 100, 96)])
def test_rope_fused_into_attention_and_append_is_bit_identical(B, T, seen, W):
    """M-RoPE folded into the SWA kernel's Q load / new-key staging and into the ring append must equal
    ivl_mrope_fwd followed by the plain kernels bit for bit (outputs and ring contents), across ring wrap, seam tiles,
    decode and split-KV shapes."""
    from infinitevl_amd import ops
    Hq, Hkv, d, C = 16, 2, 128, W - 1
    g_ = torch.Generator(device=DEV).manual_seed(T + seen)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    q, k, v = rn(B, T, Hq, d), rn(B, T, Hkv, d), rn(B, T, Hkv, d)
    kc, vc = rn(B, Hkv, C, d), rn(B, Hkv, C, d)
    pos = torch.stack([torch.arange(seen, seen + T), torch.arange(T) // 3 + seen, torch.arange(T) % 7 + seen])[:, None, :]
    pos = pos.expand(3, B, T).to(DEV)
    from infinitevl_amd.harness import InfiniteVLTextConfig
    from infinitevl_amd.modules import InfiniteVLRotaryEmbedding
    cos, sin = InfiniteVLRotaryEmbedding(InfiniteVLTextConfig())(q, pos)
    sec = [16, 24, 24]
    pos_dev = torch.full((1,), seen, dtype=torch.int64, device=DEV)
    # unfused
    q1, k1 = q.clone(), k.clone()
    ops.apply_mrope_inplace(q1, k1, cos, sin, sec)
    kc1, vc1 = kc.clone(), vc.clone()
    o1 = ops.swa_forward(q1, k1, v, window=W, scaling=d ** -0.5, k_cache=kc1, v_cache=vc1, pos_dev=pos_dev)
    ops.swa_cache_append(k1, v, kc1, vc1, pos_dev=pos_dev)
    # fused
    kc2, vc2 = kc.clone(), vc.clone()
    o2 = ops.swa_forward(q, k, v, window=W, scaling=d ** -0.5, k_cache=kc2, v_cache=vc2, pos_dev=pos_dev, rope=(cos, sin, sec))
    ops.swa_cache_append(k, v, kc2, vc2, pos_dev=pos_dev, rope=(cos, sin, sec))
    assert torch.equal(o1, o2)
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)


def test_swa_reference_vectors_d128():
    """Row S3 pinned DIRECTLY to the reference: outputs of the reference's eager_attention_forward (fp32, band mask from
    the S2 predicate) at the kernel's head shape d = 128 -- GQA group 8 and 16-over-2 heads, empty / partly / fully cached,
    decode, windows smaller and larger than the call -- fed to the operator-level entry point (concatenated K/V)."""
    from infinitevl_amd import ops
    z = load_golden("swa_attention_d128")
    for name in [str(n) for n in z["names"]]:
        q, k, v = (bf(z[name + s_]).to(DEV) for s_ in ("_q", "_k", "_v"))
        out, _ = ops.swa_attention_interface(None, q, k, v, None, scaling=128 ** -0.5, sliding_window=int(z[name + "_W"]))
        e = rms_rel(z[name + "_out"], out.float().cpu())
        assert e < 5e-3, (name, e)


# ---------------------------------------------------------------------------------------------
# vision-tower window attention
# ---------------------------------------------------------------------------------------------
def test_vision_attention_reference_vectors():
    """Pinned DIRECTLY to the reference (eager_attention_forward per window on the outputs of apply_rotary_pos_emb_vision,
    head_dim 80): windows, one full segment, ragged (an empty and a one-patch segment).  The rotary embedding folded into
    the kernel's loads must be BIT-identical to the reference's bf16 results: the fused call equals the call on the
    reference's own rotated q / k, bit for bit."""
    from infinitevl_amd import ops
    z = load_golden("vision_attention")
    q, k, v, qr, kr = (bf(z[n]).to(DEV) for n in ("q", "k", "v", "q_rot", "k_rot"))
    cos, sin = z["cos"].to(DEV), z["sin"].to(DEV)
    for name, max_len in (("windows", 64), ("full", 136), ("ragged", 69), ("ragged", 200)):
        cu = z[name + "_cu"].to(DEV)
        fused = ops.vision_window_attention(q, k, v, cu, max_len, rope=(cos, sin))
        plain = ops.vision_window_attention(qr, kr, v, cu, max_len)
        assert torch.equal(fused, plain), name
        e = rms_rel(z[name + "_out"], fused.float().cpu())
        assert e < 5e-3, (name, e)


@pytest.mark.parametrize("d,H", [(80, 16), (64, 4), (128, 2)])
def test_vision_attention_real_shapes_vs_oracle(d, H):
    """The 3B tower's shapes (16 heads x 80): a window layer (one 32 x 32-patch frame = 16 windows of 64 patches, plus edge
    windows of a 30 x 30 frame: 36 / 24 / 16 patches) and a full-attention layer (one segment per frame, 1024 and 900
    patches), q / k / v taken as strided slices of one [S, 3, H, d] projection output.  Oracle with the probabilities
    rounded to bf16 before P @ V (the kernel's rounding point): <= 5e-4... the output itself is rounded to bf16."""
    from oracle import vision
    from infinitevl_amd import ops
    torch.manual_seed(5)
    win = [64] * 16 + [64] * 9 + [48] * 6 + [36]                   # 1024 + 900 patches
    for lens in (win, [1024, 900], [1024, 900, 1000, 1024, 130]):      # the last one (d = 80) runs the 128-row workgroups
        S = sum(lens)
        cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
        qkv = bf(torch.randn(S, 3, H, d))
        pos_hw = torch.randint(0, 32, (S, 2))
        cos, sin = vision.vision_rotary_tables(pos_hw, d)
        got = ops.vision_window_attention(qkv.to(DEV)[:, 0], qkv.to(DEV)[:, 1], qkv.to(DEV)[:, 2], cu.to(DEV), max(lens),
                                          rope=(cos.to(DEV), sin.to(DEV)))
        qr, kr = vision.apply_rotary_pos_emb_vision(qkv[:, 0], qkv[:, 1], cos, sin)
        ref = vision.segment_attention(qr, kr, qkv[:, 2], cu.tolist(), p_round_dtype=torch.bfloat16)
        exact = vision.segment_attention(qr, kr, qkv[:, 2], cu.tolist())
        assert rms_rel(ref, got.float().cpu()) < 3e-3 and rms_rel(exact, got.float().cpu()) < 5e-3, (d, len(lens))


@pytest.mark.parametrize("seed", range(int(os.environ.get("IVL_VISION_FUZZ", "12"))))
def test_vision_attention_random_segmentations_vs_oracle(seed):
    """Random segmentations of the patch axis (empty, one-patch, window-sized, ragged and frame-sized segments mixed in one
    call; head_dim 80 / 64 / 128; with and without the rotary embedding in the loads; `max_seqlen` exact or generous) against
    the oracle, per row as well as over the tensor.  IVL_VISION_FUZZ=N runs N seeds (default 12)."""
    import random
    from oracle import vision
    from infinitevl_amd import ops
    r = random.Random(seed)
    d, H = r.choice([(80, 16), (80, 4), (64, 4), (128, 2)])
    kind = r.choice(["windows", "frames", "mixed"])
    pick = {"windows": lambda: r.choice([0, 1, 7, 16, 24, 36, 48, 63, 64]),
            "frames": lambda: r.choice([65, 130, 500, 900, 1024, 1100]),
            "mixed": lambda: r.choice([0, 1, 5, 64, 64, 64, 100, 129, 700, 1024])}[kind]
    lens = [pick() for _ in range(r.randint(1, 24 if kind == "windows" else 5))]
    if sum(lens) == 0:
        lens.append(9)
    S = sum(lens)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    g_ = torch.Generator().manual_seed(500 + seed)
    qkv = bf(torch.randn(S, 3, H, d, generator=g_))
    rope = r.random() < 0.7
    max_len = max(lens) + r.choice([0, 0, 3, 200])
    qd, kd, vd = (qkv.to(DEV)[:, i] for i in range(3))
    qr, kr = qkv[:, 0], qkv[:, 1]
    if rope:
        pos_hw = torch.randint(0, 32, (S, 2), generator=g_)
        cos, sin = vision.vision_rotary_tables(pos_hw, d)
        got = ops.vision_window_attention(qd, kd, vd, cu.to(DEV), max_len, rope=(cos.to(DEV), sin.to(DEV)))
        qr, kr = vision.apply_rotary_pos_emb_vision(qr, kr, cos, sin)
    else:
        got = ops.vision_window_attention(qd.contiguous(), kd.contiguous(), vd.contiguous(), cu.to(DEV), max_len)
    torch.cuda.synchronize()
    ref = vision.segment_attention(qr, kr, qkv[:, 2], cu.tolist(), p_round_dtype=torch.bfloat16)
    case = dict(d=d, H=H, lens=lens, rope=rope, max_len=max_len)
    got = got.float().cpu()
    assert torch.isfinite(got).all(), case
    assert rms_rel(ref, got) < 3e-3, (rms_rel(ref, got), case)
    row = ((ref - got) ** 2).sum(-1).sqrt() / (ref ** 2).sum(-1).sqrt().clamp_min(1e-3)
    assert float(row.max()) < 4e-2, (float(row.max()), case)


def test_vision_attention_module_matches_reference():
    """InfiniteVLVisionAttention (same parameter names as the checkpoint) against the reference module's eager output,
    hidden 320 = 4 heads x 80; the projections are bf16 GEMMs here (fp32 in the fixture)."""
    from types import SimpleNamespace
    from infinitevl_amd.modules import InfiniteVLVisionAttention
    z = load_golden("vision_attention")
    mod = InfiniteVLVisionAttention(SimpleNamespace(hidden_size=320, num_heads=4)).to(DEV, torch.bfloat16)
    mod.load_state_dict({"qkv.weight": bf(z["mod_qkv_weight"]), "qkv.bias": bf(z["mod_qkv_bias"]),
                         "proj.weight": bf(z["mod_proj_weight"]), "proj.bias": bf(z["mod_proj_bias"])})
    x = bf(z["mod_x"]).to(DEV)
    pe = (z["cos"].to(DEV), z["sin"].to(DEV))
    with torch.no_grad():
        for name in ("windows", "full"):
            cu = z[name + "_cu"].to(DEV)
            out = mod(x, cu, position_embeddings=pe)                               # max_seqlen read from cu_seqlens
            out2 = mod(x, cu, position_embeddings=pe, max_seqlen=136)              # the graph-capturable form
            assert torch.equal(out, out2)
            e = rms_rel(z[f"mod_{name}_out"], out.float().cpu())
            assert e < 1e-2, (name, e)


def test_vision_attention_is_graph_capturable_and_rejects_bad_arguments():
    from infinitevl_amd import ops
    torch.manual_seed(6)
    S, H, d = 200, 16, 80
    q, k, v = (bf(torch.randn(S, H, d)).to(DEV) for _ in range(3))
    cu = torch.tensor([0, 64, 128, 200], dtype=torch.int32, device=DEV)
    eager = ops.vision_window_attention(q, k, v, cu, 72)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.vision_window_attention(q, k, v, cu, 72)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            o = ops.vision_window_attention(q, k, v, cu, 72)
    cu.copy_(torch.tensor([0, 70, 130, 200], dtype=torch.int32))       # new segmentation, same graph (lengths within max_seqlen)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(o, ops.vision_window_attention(q, k, v, cu, 72)) and not torch.equal(o, eager)
    with pytest.raises(ValueError):                                       # head_dim not built: IVL_ERR_UNSUPPORTED
        ops.vision_window_attention(q[..., :48].contiguous(), k[..., :48].contiguous(), v[..., :48].contiguous(), cu, 72)


def test_hot_kernels_are_bit_stable_from_run_to_run():
    """Every kernel of the path is deterministic by construction (no atomics on data, fixed reduction orders), so the same
    inputs must give the same bits on every launch.  This is the test that catches what parity tolerances hide: a missing
    wait state after an MFMA (the attention kernels once read accumulators one instruction early: 1-ulp differences in
    8-15 % of the outputs from run to run), an LDS race, a DMA awaited by the wrong count."""
    from infinitevl_amd import ops
    torch.manual_seed(11)
    dev = DEV
    runs = []
    # chunk rule (pre-pass + scan, bf16 and e4m3 operands), with and without a carried state
    for T, B in ((256, 1), (1000, 2)):
        q, k = (bf(torch.randn(B, T, 16, 128)).to(dev) for _ in range(2))
        v = bf(torch.randn(B, T, 16, 256)).to(dev)
        g = torch.nn.functional.logsigmoid(torch.randn(B, T, 16)).to(dev)
        beta = bf(torch.rand(B, T, 16)).to(dev)
        h0 = torch.randn(B, 16, 128, 256).to(dev)
        for mma in ("bf16", "fp8_e4m3"):
            runs.append((f"gdn_chunk T={T} {mma}", lambda q=q, k=k, v=v, g=g, beta=beta, h0=h0, mma=mma: ops.chunk_gated_delta_rule(
                q, k, v, g, beta, initial_state=h0, output_final_state=True, use_qk_l2norm_in_kernel=True, mma_dtype=mma)))
    # attention: split-KV step over a full ring, one-split long causal call, decode row
    W = 4096
    kc, vc = (bf(torch.randn(1, 2, W - 1, 128)).to(dev) for _ in range(2))
    pos = torch.full((1,), 3 * W + 17, dtype=torch.int64, device=dev)
    for T in (256, 1):
        qs = bf(torch.randn(1, T, 16, 128)).to(dev)
        kn, vn = (bf(torch.randn(1, T, 2, 128)).to(dev) for _ in range(2))
        runs.append((f"swa ring T={T}", lambda qs=qs, kn=kn, vn=vn: (ops.swa_forward(qs, kn, vn, window=W, scaling=128 ** -0.5,
                                                                                    k_cache=kc, v_cache=vc, pos_dev=pos),)))
    qs = bf(torch.randn(1, 2048, 16, 128)).to(dev)
    kn, vn = (bf(torch.randn(1, 2048, 2, 128)).to(dev) for _ in range(2))
    runs.append(("swa causal T=2048", lambda: (ops.swa_forward(qs, kn, vn, window=8192, scaling=128 ** -0.5),)))
    # vision attention: window layer, full layer (128-row workgroups + key pre-pass)
    qkv = bf(torch.randn(4096, 3, 16, 80)).to(dev)
    tabs = tuple(torch.randn(4096, 80, device=dev) for _ in range(2))
    for seg in (64, 1024):
        cu = torch.arange(0, 4097, seg, dtype=torch.int32, device=dev)
        runs.append((f"vision seg={seg}", lambda cu=cu, seg=seg: (ops.vision_window_attention(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, seg,
                                                                                             rope=tabs),)))
    for name, fn in runs:
        first = [t.clone() for t in fn() if t is not None]
        for rep in range(6):
            again = [t for t in fn() if t is not None]
            for a_, b_ in zip(first, again):
                assert torch.equal(a_, b_), (name, rep)


def _band_counts(n_prev, T, W):
    """Decode WHICH keys each row attended: q = 0 makes the softmax uniform over the visible set, V holds
    one-hot residues of the key index, so out[i, r] * n_vis(i) = #visible keys with residue r.  Two encodings: j % 128
    (fine position) and (j // 32) % 128 (coarse position); at most 34 keys share a class, so the count survives the bf16
    roundings of the output and of the split-KV partial rows (count * 2^-8 < 0.14) - a contiguous band [lo, hi] is pinned
    by the two histograms."""
    lo, hi = oswa.window_bounds(n_prev, T, W)
    S = n_prev + T
    j = np.arange(S)
    exp_lo = np.stack([np.bincount(j[lo[i]:hi[i] + 1] % 128, minlength=128) for i in range(T)])
    exp_hi = np.stack([np.bincount((j[lo[i]:hi[i] + 1] // 32) % 128, minlength=128) for i in range(T)])
    return lo, hi, exp_lo, exp_hi


@pytest.mark.parametrize("W,seen,T", [(8, 0, 5), (8, 0, 19), (8, 7, 6), (8, 40, 1), (96, 250, 70), (4096, 0, 300),
                                      (4096, 4000, 256), (4096, 9000, 256), (4096, 9000, 1), (4096, 4095, 64),
                                      (4096, 0, 8192), (1024, 5000, 8192)])   # last two: 128-row workgroups
def test_swa_band_indices_bit_exact(W, seen, T):
    """The integer contract: the set of keys each query row sees."""
    from infinitevl_amd import ops
    Hq, Hkv, d = 16, 2, 128
    n_prev = oswa.n_prev_keys(W, seen)
    lo, hi, exp_lo, exp_hi = _band_counts(n_prev, T, W)
    S = n_prev + T
    q = torch.zeros(1, Hq, T, d, dtype=torch.bfloat16, device=DEV)
    k = bf(torch.randn(1, Hkv, S, d)).to(DEV)
    j = torch.arange(S)
    for which, expect in (("lo", exp_lo), ("hi", exp_hi)):
        idx = (j % 128) if which == "lo" else (j // 32) % 128
        v = torch.nn.functional.one_hot(idx, 128).to(torch.bfloat16)[None, None].expand(1, Hkv, S, d).contiguous().to(DEV)
        out, _ = ops.swa_attention_interface(None, q, k, v, None, scaling=d ** -0.5, sliding_window=W)
        got = out.float().cpu()[0]                                  # [T, Hq, 128]
        n_vis = torch.from_numpy(hi - lo + 1).float()[:, None, None]
        counts = torch.round(got * n_vis).to(torch.int64)
        for h in (0, 7, 15):
            assert np.array_equal(counts[:, h].numpy(), expect), (which, h)


def test_swa_128_row_workgroups_equal_64_row_workgroups():
    """Batched prefill call == the per-sequence calls bit for bit (a row's tiles and their order do not depend on the
    grid: same 128-row workgroups of swa_prefill_kernel, different block order / XCD placement), ring wrap included."""
    from infinitevl_amd import ops
    B, T, Hq, Hkv, d, W = 2, 4096, 16, 2, 128, 1024
    C = W - 1
    torch.manual_seed(11)
    kc, vc = bf(torch.randn(B, Hkv, C, d)).to(DEV), bf(torch.randn(B, Hkv, C, d)).to(DEV)
    pos_dev = torch.full((1,), 3000, dtype=torch.int64, device=DEV)
    q, kn, vn = (bf(torch.randn(B, T, h, d)).to(DEV) for h in (Hq, Hkv, Hkv))
    o_big = ops.swa_forward(q, kn, vn, window=W, scaling=d ** -0.5, k_cache=kc, v_cache=vc, pos_dev=pos_dev)
    for b in range(B):
        o_one = ops.swa_forward(q[b:b + 1], kn[b:b + 1], vn[b:b + 1], window=W, scaling=d ** -0.5,
                                k_cache=kc[b:b + 1].contiguous(), v_cache=vc[b:b + 1].contiguous(), pos_dev=pos_dev)
        assert torch.equal(o_big[b:b + 1], o_one), b
    assert torch.isfinite(o_big.float()).all()


def test_swa_ring_path_equals_concatenated_path():
    """Full-size property (InfiniteVL-3B heads, W=4096, T=256): attention over (ring cache ++ new) ==
    attention over the torch.cat'ed tensors (same keys through two data paths; only the split-KV
    summation order may differ -> one bf16 ulp), and the ring content equals the reference tail bit for bit."""
    from infinitevl_amd import ops
    B, T, Hq, Hkv, d, W = 1, 256, 16, 2, 128, 4096
    C = W - 1
    torch.manual_seed(3)
    kc = torch.zeros(B, Hkv, C, d, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    pos_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
    hist_k, hist_v = [], []
    for step in range(20):                                         # 5120 tokens: the ring wraps
        kn, vn = bf(torch.randn(B, T, Hkv, d)).to(DEV), bf(torch.randn(B, T, Hkv, d)).to(DEV)
        qn = bf(torch.randn(B, T, Hq, d)).to(DEV)
        o_ring = ops.swa_forward(qn, kn, vn, window=W, scaling=d ** -0.5, k_cache=kc, v_cache=vc, pos_dev=pos_dev)
        hist_k.append(kn)
        hist_v.append(vn)
        k_all, v_all = torch.cat(hist_k, 1), torch.cat(hist_v, 1)
        n_prev = min(C, step * T)
        o_cat = ops.swa_forward(qn, k_all[:, k_all.shape[1] - n_prev - T:], v_all[:, v_all.shape[1] - n_prev - T:],
                                window=W, scaling=d ** -0.5)
        assert rms_rel(o_cat.float().cpu(), o_ring.float().cpu()) < 4e-3, step
        ops.swa_cache_append(kn, vn, kc, vc, pos_dev=pos_dev)
        ops.counter_add(pos_dev, T)
    assert int(pos_dev.item()) == 20 * T
    tail = torch.cat(hist_k, 1)[:, -C:]                            # chronological last W-1 keys
    idx = (torch.arange(C, device=DEV) + (20 * T - C)) % C
    assert torch.equal(kc.index_select(2, idx).transpose(1, 2), tail)


# ---------------------------------------------------------------------------------------------
# layers, harness (row H), hipGraph, clone, constant memory
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fuse", [False, True])
def test_layer_stack_vs_oracle(fuse):
    """fuse=False: one kernel per reference operator (operator-level drop-in);
    fuse=True: fused projections + prologue/epilogue kernels (module-level fast path)."""
    r = parity.layer_parity(DEV, T_prefill=130, n_decode=3, window=96, seed=0, fuse=fuse)
    for name in ("prefill", "stream", "decode0", "decode1", "decode2"):
        assert r[name] < 1.5e-2, r
    assert r["gdn_state"] < 1.5e-2 and r["swa_keys"] < 6e-3, r


@pytest.mark.parametrize("seed", range(int(os.environ.get("IVL_SCHED_FUZZ", "6"))))
def test_layer_stack_random_call_schedules_vs_oracle(seed):
    """A 4-layer stack (1 sliding + 3 linear layers, real head shapes) fed a RANDOM sequence of call lengths over one cache
    -- 1-, 2- and 3-token calls (shorter than the convolution's carry), the 64 / 65 mode switch (Q7), ragged chunks, calls
    longer than the window, ring wraps in the middle of a call -- against the oracle with the reference's rounding points,
    call by call, and the cache at the end.  IVL_SCHED_FUZZ=N runs N seeds (default 6)."""
    import random
    r = random.Random(seed)
    window = r.choice([64, 96, 300])
    fuse = r.random() < 0.6
    n = r.randint(5, 9)
    schedule = [r.choice([1, 1, 2, 3, 5, 17, 63, 64, 65, 70, 130, 256, 333]) for _ in range(n)]
    res = parity.layer_parity(DEV, window=window, seed=100 + seed, fuse=fuse, schedule=schedule)
    case = dict(window=window, fuse=fuse, schedule=schedule)
    for i in range(n):
        assert res[f"call{i}"] < 1.5e-2, (res, case)
    assert res["gdn_state"] < 1.5e-2 and res["swa_keys"] < 6e-3, (res, case)


def _small_stack(window=96, seed=3, fuse=True):
    from infinitevl_amd.harness import InfiniteVLTextStack
    from oracle import model as omodel
    hc, oc = parity.small_configs(window)
    params = parity.bf16_params(omodel.random_params(oc, seed=seed, vocab=hc.vocab_size))
    stack = InfiniteVLTextStack(hc)
    parity.load_params(stack, params)
    stack = stack.to(DEV, torch.bfloat16).eval()
    if fuse:
        stack.fuse_()
    return stack, hc, oc, params


def test_full_width_period_vs_oracle():
    """One 4-layer period at the model's real width (hidden 2048, 16 GDN heads, 16/2 SWA heads, d=128/256), fused
    fast path: 256-token prefill, a 100-token streaming frame over a wrapping 200-key window, decode steps.  At this
    width bf16 rounding alone moves the 4-layer output by ~1.7e-2 (oracle with the reference's rounding points vs the
    oracle in exact fp32), so the bound is stated against that noise floor: the HIP path is no farther from the exact
    result than 1.25x the reference-rounding model is, and within 2.5e-2 of that model."""
    r = parity.layer_parity(DEV, T_prefill=256, n_decode=2, window=200, seed=3, stream_T=100, fuse=True, heads=16,
                            noise_floor=True)
    for name in ("prefill", "stream", "decode0", "decode1"):
        assert r[name] < 2.5e-2, r
        assert r[name + "_vs_exact"] < 1.25 * r[name + "_bf16model_vs_exact"] + 1e-3, r
    assert r["gdn_state"] < 2e-2 and r["swa_keys"] < 5e-3, r


def test_fused_kernels_equal_unfused_kernels():
    """The fused prologue/epilogue kernels keep the rounding points of the single-purpose ones: conv outputs,
    states, gates, rope and norms are bit-identical; only the GEMM fusion may reorder fp32 accumulation."""
    from infinitevl_amd import ops
    torch.manual_seed(11)
    B, T, H, K, V = 2, 37, 4, 128, 256
    Dq, Dk, Dv = H * K, H * K, H * V
    cols = (0, Dq, Dq + Dk, Dq + Dk + Dv + H * V, Dq + Dk + Dv + H * V + H)
    ld = (cols[4] + H + 7) // 8 * 8
    proj = bf(torch.randn(B, T, ld)).to(DEV)
    ws = [bf(torch.randn(D, 1, 4) * 0.4).to(DEV) for D in (Dq, Dk, Dv)]
    st = [bf(torch.randn(B, D, 4)).to(DEV) for D in (Dq, Dk, Dv)]
    A_log, dt = torch.randn(H).to(DEV), torch.randn(H).to(DEV)
    st_f = [s_.clone() for s_ in st]
    q, k, v, g, beta = ops.gdn_prologue(proj, cols, ws, st_f, st_f, A_log, dt, H, Dq, Dk, Dv)
    outs, sts = [], []
    for i, (c0, D) in enumerate(zip(cols[:3], (Dq, Dk, Dv))):
        conv = ops.ShortConvolution(D, 4).to(DEV, torch.bfloat16)
        with torch.no_grad():
            conv.weight.copy_(ws[i])
        s_i = st[i].clone()
        y, s_i = conv(proj[..., c0:c0 + D].contiguous(), cache=s_i, output_final_state=True)
        outs.append(y)
        sts.append(s_i)
    g2, b2 = ops.gdn_gate(proj[..., cols[3]:cols[3] + H].contiguous(), proj[..., cols[4]:cols[4] + H].contiguous(), A_log, dt)
    for a_, b_ in zip((q, k, v), outs):
        assert torch.equal(a_, b_)
    for a_, b_ in zip(st_f, sts):
        assert torch.equal(a_, b_)
    assert torch.equal(g, g2) and torch.equal(beta, b2)
    # gated norm with the gate read in place
    o = bf(torch.randn(B, T, H, V)).to(DEV)
    norm = ops.FusedRMSNormGated(V).to(DEV, torch.bfloat16)
    cg = Dq + Dk + Dv
    y1 = norm(o, proj[..., cg:cg + H * V].reshape(B, T, H, V).contiguous())
    y2 = ops.rmsnorm_swish_gate_strided(o, proj[..., cg:], ld, norm.weight, norm.eps)
    assert torch.equal(y1, y2)
    # add + rmsnorm vs torch bf16 eager (Qwen2RMSNorm arithmetic)
    x, r = bf(torch.randn(3, 50, 2048)).to(DEV), bf(torch.randn(3, 50, 2048)).to(DEV)
    w = bf(1 + 0.1 * torch.randn(2048)).to(DEV)
    y, h = ops.add_rmsnorm(x, r, w, 1e-6)
    h_ref = r + x
    xf = h_ref.float()
    y_ref = w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16)
    assert torch.equal(h, h_ref) and rms_rel(y_ref.float().cpu(), y.float().cpu()) < 2e-3
    # SwiGLU gate vs torch bf16 eager
    gu = bf(torch.randn(5, 7, 2 * 1376)).to(DEV)
    ref = torch.nn.functional.silu(gu[..., :1376]) * gu[..., 1376:]
    assert torch.equal(ops.silu_mul(gu), ref)
    # strided M-RoPE vs contiguous M-RoPE
    qkv = bf(torch.randn(2, 9, (4 + 2 + 2) * 128)).to(DEV)
    pos = torch.arange(9)[None, None, :].expand(3, 2, 9).contiguous()
    cos, sin = oswa.rotary_cos_sin(pos, 128, 1e6)
    qc = qkv[..., :512].reshape(2, 9, 4, 128).contiguous()
    kc = qkv[..., 512:768].reshape(2, 9, 2, 128).contiguous()
    ops.apply_mrope_inplace(qc, kc, bf(cos).to(DEV), bf(sin).to(DEV), [16, 24, 24])
    qv = qkv[..., :512].unflatten(-1, (4, 128))
    kv = qkv[..., 512:768].unflatten(-1, (2, 128))
    ops.apply_mrope_strided_inplace(qv, kv, bf(cos).to(DEV), bf(sin).to(DEV), [16, 24, 24])
    assert torch.equal(qv, qc) and torch.equal(kv, kc)


def test_hipgraph_step_is_bit_exact_with_eager_and_fixes_frozen_window():
    """One captured graph serves every step (device-resident counters): outputs, state and counters equal
    the eager run bit for bit over 8 replays that cross the window fill."""
    from infinitevl_amd.harness import GraphedStep
    stack, hc, _, _ = _small_stack(window=200)
    T = 70
    xs = [bf(torch.randn(1, T, hc.hidden_size) * 0.5).to(DEV) for _ in range(9)]
    with torch.no_grad():
        c1 = stack.allocate_inference_cache(1)
        eager, pos = [], 0
        for x in xs:
            pid = torch.arange(pos, pos + T, device=DEV)[None, None, :].expand(3, 1, T)
            eager.append(stack(inputs_embeds=x, position_ids=pid, past_key_values=c1)[0].clone())
            pos += T
        c2 = stack.allocate_inference_cache(1)
        pid = torch.arange(0, T, device=DEV)[None, None, :].expand(3, 1, T)
        stack(inputs_embeds=xs[0], position_ids=pid, past_key_values=c2)
        gs = GraphedStep(stack, c2, 1, T)
        for i in range(1, 9):
            h, _ = gs.step(xs[i])
            assert torch.equal(eager[i], h), i
    assert c2.layers[0].cumulative_length == c1.layers[0].cumulative_length == 9 * T
    assert c2.layers[0].size == c1.layers[0].size == 199
    assert int(c2.layers[0]._pos_dev.item()) == 9 * T
    assert torch.equal(c1.layers[1].recurrent_state, c2.layers[1].recurrent_state)
    assert torch.equal(c1.layers[0].keys, c2.layers[0].keys)


def test_hipgraph_from_a_fresh_poisoned_cache_equals_eager_first_call():
    """The reference's first call ignores whatever the (torch.empty) cache tensors hold (std:298-300).
    A replayed graph always reads them, so capture must zero them first: poison the cache with NaN and
    compare against the eager path from step 0."""
    from infinitevl_amd.harness import GraphedStep
    stack, hc, _, _ = _small_stack(window=96)
    T = 70
    xs = [bf(torch.randn(1, T, hc.hidden_size) * 0.5).to(DEV) for _ in range(3)]
    with torch.no_grad():
        c1, c2 = stack.allocate_inference_cache(1), stack.allocate_inference_cache(1)
        for c in (c1, c2):
            for layer in c.layers:
                for t in vars(layer).values():
                    if torch.is_tensor(t) and t.is_floating_point():
                        t.fill_(float("nan"))
        gs = GraphedStep(stack, c2, 1, T)
        pos = 0
        for x in xs:
            pid = torch.arange(pos, pos + T, device=DEV)[None, None, :].expand(3, 1, T)
            h1 = stack(inputs_embeds=x, position_ids=pid, past_key_values=c1)[0]
            h2, _ = gs.step(x)
            assert torch.isfinite(h1.float()).all() and torch.equal(h1, h2)
            pos += T


def test_graph_survives_a_larger_eager_call_and_a_reset():
    """(a) the scratch buffer baked into a captured graph must stay alive when a later eager call outgrows
    it (retired, not freed); (b) reset() + replay of an existing graph equals a fresh eager sequence."""
    from infinitevl_amd import ops
    from infinitevl_amd.harness import GraphedStep
    stack, hc, _, _ = _small_stack(window=96)
    T = 70
    xs = [bf(torch.randn(1, T, hc.hidden_size) * 0.5).to(DEV) for _ in range(4)]
    torch.cuda.synchronize()
    ops._WORKSPACES.clear()              # start from small buffers whatever ran before (no graph of an earlier test is alive)
    ops._GRAPH_PINNED.clear()
    with torch.no_grad():
        c1, c2 = stack.allocate_inference_cache(1), stack.allocate_inference_cache(1)
        gs = GraphedStep(stack, c2, 1, T)
        gs.capture()
        ws_before = {k: v.data_ptr() for k, v in ops._WORKSPACES.items()}
        # an eager call far larger than the frame: both the GDN records and the SWA partials outgrow their buffers
        big = bf(torch.randn(1, 4200, hc.hidden_size) * 0.5).to(DEV)
        cbig = stack.allocate_inference_cache(1)
        stack(inputs_embeds=big, position_ids=torch.arange(4200, device=DEV)[None, None, :].expand(3, 1, 4200),
              past_key_values=cbig)
        grown = [k for k, v in ops._WORKSPACES.items() if k in ws_before and v.data_ptr() != ws_before[k]]
        assert grown, "the big call was meant to outgrow a captured workspace"
        assert len(ops._RETIRED) >= len(grown)
        junk = [torch.full((1 << 22,), 7, dtype=torch.uint8, device=DEV) for _ in range(8)]     # would land on freed memory
        pos = 0
        for x in xs[:2]:
            pid = torch.arange(pos, pos + T, device=DEV)[None, None, :].expand(3, 1, T)
            h1 = stack(inputs_embeds=x, position_ids=pid, past_key_values=c1)[0]
            h2, _ = gs.step(x)
            assert torch.equal(h1, h2)
            pos += T
        del junk
        # (b) new sequence on the same graph
        gs.reset()
        c3 = stack.allocate_inference_cache(1)
        pos = 0
        for x in xs[2:]:
            pid = torch.arange(pos, pos + T, device=DEV)[None, None, :].expand(3, 1, T)
            h1 = stack(inputs_embeds=x, position_ids=pid, past_key_values=c3)[0]
            h2, _ = gs.step(x)
            assert torch.equal(h1, h2)
            pos += T
        assert c2.get_seq_length() == 2 * T


def test_swa_crop_keeps_the_last_tokens_like_the_reference():
    """std:192-213: crop keeps the LAST new_size tokens at the front and restarts the counters."""
    from infinitevl_amd.cache import StaticSlidingWindowLayerPrealloc
    from infinitevl_amd.harness import InfiniteVLTextConfig
    cfg = InfiniteVLTextConfig(sliding_window=64, num_hidden_layers=4)
    layer = StaticSlidingWindowLayerPrealloc(config=cfg, batch_size=1, device=DEV, dtype=torch.bfloat16)
    k = bf(torch.randn(1, 2, 20, 128)).to(DEV)
    v = bf(torch.randn(1, 2, 20, 128)).to(DEV)
    layer.update(k, v)
    layer.crop(7)
    assert layer.size == 7 and layer.cumulative_length == 7 and int(layer._pos_dev.item()) == 7
    assert torch.equal(layer.keys, k[:, :, 13:20]) and torch.equal(layer.values, v[:, :, 13:20])
    layer.crop(-3)
    assert layer.size == 4 and torch.equal(layer.keys, k[:, :, 16:20])
    layer.update(bf(torch.randn(1, 2, 70, 128)).to(DEV), bf(torch.randn(1, 2, 70, 128)).to(DEV))
    with pytest.raises(ValueError):
        layer.crop(3)


def test_stack_realdims_reference_vectors():
    """Rows S0, G0 and H pinned DIRECTLY to the reference: `stack_realdims.npz` holds what the reference's own
    InfiniteVLDecoderLayer / GatedDeltaNet / InfiniteVLSelfAttention / StaticCachePrealloc produce (fp32) at the head
    shapes the kernels are built for, on bf16-representable weights and inputs; the drop-in modules run the same
    scenario in bf16: prefill 70 -> two 20-token frames (the 96-key window wraps) -> cache clone -> greedy decode 6 ->
    one more frame on the original.  Tolerance: bf16 activations through 2 decoder layers at width 256 sit ~1e-2 from the
    fp32 reference (hidden states: 3e-2 bound); greedy tokens must agree wherever the reference's top-2 logit margin
    exceeds 10x the logit error."""
    from infinitevl_amd.harness import (InfiniteVLTextConfig, InfiniteVLTextStack, clone_inference_cache, greedy_decode)
    z = load_golden("stack_realdims")
    lt = [str(x) for x in z["layer_types"]]
    hc = InfiniteVLTextConfig(vocab_size=97, hidden_size=256, intermediate_size=128, num_hidden_layers=2,
                              num_attention_heads=2, num_key_value_heads=1, head_dim=128, sliding_window=96, layer_types=lt,
                              num_linear_heads=2, num_linear_key_value_heads=2, linear_head_dim=128, rope_theta=1e6)
    params = {k[2:]: v for k, v in z.items() if k.startswith("w.")}
    for fuse in (False, True):
        stack = InfiniteVLTextStack(hc)
        parity.load_params(stack, params)
        stack = stack.to(DEV, torch.bfloat16).eval()
        if fuse:
            stack.fuse_()
        pid = lambda s0, T: torch.arange(s0, s0 + T, device=DEV)[None, None, :].expand(3, 1, T)    # noqa: E731
        with torch.no_grad():
            cache = stack.allocate_inference_cache(1)
            h0, _ = stack(input_ids=z["s.ids0"].to(DEV), position_ids=pid(0, 70), past_key_values=cache)
            assert rms_rel(z["s.h0"], h0.float().cpu()) < 3e-2, rms_rel(z["s.h0"], h0.float().cpu())
            pos = 70
            for f in range(2):
                h, lg = stack(inputs_embeds=bf(z["s.frames"][f]).to(DEV), position_ids=pid(pos, 20), past_key_values=cache)
                assert rms_rel(z["s.h_frames"][f], h.float().cpu()) < 3e-2, f
                pos += 20
            qa = clone_inference_cache(cache)
            embed = params["embed_tokens.weight"]
            tok, qpos = int(z["s.first_token"]), pos
            for step in range(6):
                _, lg = stack(input_ids=torch.tensor([[tok]], device=DEV), position_ids=pid(qpos, 1), past_key_values=qa)
                ref_lg = z["s.decode_logits"][step]
                err = float((lg[0, -1].float().cpu() - ref_lg).abs().max())
                top2 = torch.topk(ref_lg, 2).values
                if float(top2[0] - top2[1]) > 10 * err:
                    assert int(lg[0, -1].argmax()) == int(z["s.decode_tokens"][step]), step
                assert rms_rel(ref_lg, lg[0, -1].float().cpu()) < 5e-2, step
                tok = int(z["s.decode_tokens"][step])          # teacher forcing: follow the reference's branch
                qpos += 1
            h3, _ = stack(inputs_embeds=bf(z["s.frames"][2]).to(DEV), position_ids=pid(pos, 20), past_key_values=cache)
            assert rms_rel(z["s.h_frame2_after_clone"], h3.float().cpu()) < 3e-2
            sw = cache.layers[0]
            assert sw.size == int(z["s.swa_size"]) and sw.cumulative_length == int(z["s.swa_cum"])
            assert rms_rel(z["s.swa_keys"], sw.keys.float().cpu()) < 2e-2
            assert rms_rel(z["s.swa_values"], sw.values.float().cpu()) < 2e-2
            assert rms_rel(z["s.gdn1_recurrent"], cache.layers[1].recurrent_state.float().cpu()) < 3e-2
            # single mixers at the module boundary, no cache
            og, _ = stack.layers[1].self_attn(bf(z["m.x_gdn"]).to(DEV), past_key_values=None)
            assert rms_rel(z["m.o_gdn"], og.float().cpu()) < 2e-2, rms_rel(z["m.o_gdn"], og.float().cpu())
            xs = bf(z["m.x_swa"]).to(DEV)
            pos3 = z["m.pos3"].to(DEV)
            pe = stack.rotary_emb(xs, pos3)
            osw, _ = stack.layers[0].self_attn(xs, attention_mask=None, position_ids=pos3, past_key_values=None,
                                               cache_position=torch.arange(130, device=DEV), position_embeddings=pe)
            assert rms_rel(z["m.o_swa"], osw.float().cpu()) < 2e-2, rms_rel(z["m.o_swa"], osw.float().cpu())


def test_configs0_plumbing_hip_stack_vs_cpu_path():
    """BASELINE.json configs[0] on the HIP stack: the mllm_demo sample-0 shaped workload (64 random ids + 256 image
    placeholder embeds, 3-D M-RoPE positions) through GatedDeltaNet / InfiniteVLSelfAttention / StaticCachePrealloc with
    the infinitevl_standard signatures (one 4-layer period at InfiniteVL-3B's head shapes), greedy 16 tokens; the CPU
    path (oracle, bf16 rounding model, teacher-forced on the HIP tokens) must produce the same token wherever its top-2
    logit margin is unambiguous (more than twice the largest logit difference between the two paths: then no flip is
    possible).  A 128-entry vocabulary keeps such margins frequent with random-init weights: among 151936 near-flat
    logits no argmax would be unambiguous."""
    from infinitevl_amd.harness import InfiniteVLTextConfig, InfiniteVLTextStack
    from oracle import model as omodel
    lt = ["sliding_attention", "linear_attention", "linear_attention", "linear_attention"]
    oc = omodel.OracleConfig(hidden_size=2048, intermediate_size=2048, num_attention_heads=16, num_key_value_heads=2,
                             num_linear_heads=16, linear_head_dim=128, expand_v=2.0, conv_size=4, sliding_window=4096,
                             rope_theta=1e6, mrope_section=[16, 24, 24], layer_types=lt)
    hc = InfiniteVLTextConfig(vocab_size=128, hidden_size=2048, intermediate_size=2048, num_hidden_layers=4,
                              sliding_window=4096, layer_types=lt)
    params = parity.bf16_params(omodel.random_params(oc, seed=0, vocab=128))
    stack = InfiniteVLTextStack(hc)
    parity.load_params(stack, params)
    stack = stack.to(DEV, torch.bfloat16).eval().fuse_()
    ids, img, is_image, pos3, nxt = parity.configs0_workload(2048, 128)
    with torch.no_grad():
        cache = stack.allocate_inference_cache(1)
        x = parity.configs0_embeds(stack.embed_tokens.weight, ids, img, is_image)
        _, lg = stack(inputs_embeds=x, position_ids=pos3.to(DEV), past_key_values=cache)
        toks, logits = [], []
        for step in range(16):
            logits.append(lg[0, -1].float().cpu())
            tok = int(lg[0, -1].argmax())
            toks.append(tok)
            pid = torch.full((3, 1, 1), nxt + step, dtype=torch.int64, device=DEV)
            _, lg = stack(input_ids=torch.tensor([[tok]], device=DEV), position_ids=pid, past_key_values=cache)
    assert cache.get_seq_length() == 336 and cache.layers[0].size == 336
    assert all(torch.isfinite(l).all() for l in logits)
    otoks, ologits, _ = parity.configs0_oracle_run(oc, params, act_dtype=torch.bfloat16, kernel_rounding=torch.bfloat16,
                                                   forced_tokens=toks)
    checked, info = 0, []
    for step in range(16):
        err = float((ologits[step] - logits[step]).abs().max())
        top2 = torch.topk(ologits[step], 2).values
        info.append((round(float(top2[0] - top2[1]), 4), round(err, 4), otoks[step] == toks[step]))
        if float(top2[0] - top2[1]) > 2 * err:      # the other path cannot flip the argmax: 2 x max|dlogit| < margin
            assert otoks[step] == toks[step], (step, info)
            checked += 1
        assert rms_rel(ologits[step], logits[step]) < 5e-2, (step, rms_rel(ologits[step], logits[step]))
    print("configs0 (margin, max|dlogit|, same token):", info)
    assert checked >= 3, info


def test_clone_branch_decode_then_resume_stream():
    """demo:357-438: clone the stream cache, greedy-decode on the clone, then continue the stream on the
    original as if nothing happened; greedy tokens equal the oracle's."""
    from infinitevl_amd.harness import clone_inference_cache, greedy_decode
    from oracle import model as omodel
    stack, hc, oc, params = _small_stack(window=96)
    embed = params["embed_tokens.weight"]
    with torch.no_grad():
        cache, ocache = stack.allocate_inference_cache(1), omodel.new_cache(oc, torch.bfloat16)
        x = bf(torch.randn(1, 130, hc.hidden_size) * 0.5)
        pid = torch.arange(130)[None, None, :].expand(3, 1, 130).contiguous()
        _, logits = stack(inputs_embeds=x.to(DEV), position_ids=pid.to(DEV), past_key_values=cache)
        h_ref = omodel.text_stack(params, x.float(), pid, oc, ocache, torch.bfloat16, torch.bfloat16)
        ref_logits = h_ref[0, -1] @ embed.T
        assert rms_rel(ref_logits, logits[0, -1].float().cpu()) < 2e-2
        qa = clone_inference_cache(cache)
        first = logits[:, -1].argmax(-1)
        toks = greedy_decode(stack, qa, first, steps=4)
        # oracle greedy on a cloned oracle cache, teacher-forced with OUR tokens to avoid argmax ties
        oqa = omodel.clone_cache(ocache)
        tok, pos = int(first), 130
        for s_ in range(4):
            hq = omodel.text_stack(params, embed[torch.tensor([[tok]])], torch.full((3, 1, 1), pos), oc, oqa,
                                   torch.bfloat16, torch.bfloat16)
            lg = hq[0, -1] @ embed.T
            top2 = lg.topk(2).values
            if float(top2[0] - top2[1]) > 0.05 * float(lg.std()):      # unambiguous argmax
                assert int(lg.argmax()) == int(toks[0, s_]), s_
            tok, pos = int(toks[0, s_]), pos + 1
        assert qa.get_seq_length() == 134 and cache.get_seq_length() == 130
        x2 = bf(torch.randn(1, 70, hc.hidden_size) * 0.5)
        pid2 = torch.arange(130, 200)[None, None, :].expand(3, 1, 70).contiguous()
        h2, _ = stack(inputs_embeds=x2.to(DEV), position_ids=pid2.to(DEV), past_key_values=cache)
        h2_ref = omodel.text_stack(params, x2.float(), pid2, oc, ocache, torch.bfloat16, torch.bfloat16)
        assert rms_rel(h2_ref, h2.float().cpu()) < 1.5e-2


def test_constant_memory_over_a_long_stream():
    """The cache is the long-context mechanism: allocated memory is flat once the
    window is full, however many tokens stream through (here 80 x 256 = 20480 tokens, window 1024)."""
    from infinitevl_amd.harness import GraphedStep
    stack, hc, _, _ = _small_stack(window=1024)
    with torch.no_grad():
        cache = stack.allocate_inference_cache(1)
        gs = GraphedStep(stack, cache, 1, 256, logits_to_keep=1)
        x = bf(torch.randn(1, 256, hc.hidden_size) * 0.5).to(DEV)
        import gc
        for _ in range(8):
            gs.step(x)
        torch.cuda.synchronize()
        gc.collect()
        base = torch.cuda.memory_allocated()
        peak = base
        for _ in range(72):
            gs.step(x)
            peak = max(peak, torch.cuda.memory_allocated())
        torch.cuda.synchronize()
        assert peak <= base, (peak, base, peak - base)
        assert cache.get_seq_length() == 80 * 256 and torch.isfinite(gs.hidden.float()).all()


def test_full_size_model_streams_512k_tokens_in_constant_memory():
    """BASELINE.json configs[3]/north_star: "constant memory verified to 512K tokens".  The real InfiniteVL-3B
    decoder shape (36 layers, hidden 2048, 16/2 heads, GDN state 16x128x256, window 4096; random-init bf16),
    one sequence, 2048 hipGraph steps x 256 tokens = 524,288 tokens: device memory does not grow by a byte
    after the window is full, the device/host counters agree, and activations stay finite."""
    import gc
    from infinitevl_amd.harness import GraphedStep, InfiniteVLTextConfig, InfiniteVLTextStack
    cfg = InfiniteVLTextConfig(sliding_window=4096)
    with torch.device(DEV):
        torch.set_default_dtype(torch.bfloat16)
        try:
            model = InfiniteVLTextStack(cfg)
        finally:
            torch.set_default_dtype(torch.float32)
    model = model.to(torch.bfloat16).eval()
    model.init_weights_(seed=0).fuse_()
    with torch.no_grad():
        cache = model.allocate_inference_cache(1)
        gs = GraphedStep(model, cache, 1, 256, logits_to_keep=1)
        g_ = torch.Generator(device=DEV).manual_seed(7)
        frames = [(torch.randn(1, 256, cfg.hidden_size, device=DEV, generator=g_) * 0.02).to(torch.bfloat16) for _ in range(3)]
        for i in range(20):                      # 5120 tokens: window (4095 keys) full
            gs.step(frames[i % 3])
        torch.cuda.synchronize()
        gc.collect()
        base, peak = torch.cuda.memory_allocated(), 0
        for i in range(20, 2048):
            gs.step(frames[i % 3])
            if i % 256 == 0:
                peak = max(peak, torch.cuda.memory_allocated())
        torch.cuda.synchronize()
        peak = max(peak, torch.cuda.memory_allocated())
        assert peak <= base, (peak, base)
        assert cache.get_seq_length() == 2048 * 256 == 524288
        assert int(cache.layers[0]._pos_dev.item()) == 524288 and cache.layers[0].size == 4095
        assert torch.isfinite(gs.hidden.float()).all() and torch.isfinite(gs.logits.float()).all()
        assert cache.memory_bytes() < 80 * 2 ** 20          # 27 MiB GDN state + 1.7 MiB conv + 37.7 MB SWA ring
    del model, gs, cache
    gc.collect()
    torch.cuda.empty_cache()


def test_full_size_model_streams_1m_tokens_in_fp8():
    """BASELINE.json configs[4]: "fp8 MFMA Gated DeltaNet + SWA decode, continuous 1M-token stream, 1 MI355X".  The real
    InfiniteVL-3B decoder shape, e4m3 operands in the GDN chunk scan (every 256-token frame) and in the SWA decode step,
    4096 hipGraph steps x 256 tokens = 1,048,576 tokens with a greedy decode burst every 512 frames (as the demo answers
    questions during a stream, on a cloned cache): allocated memory flat once the window is full, finite activations to
    the end, step time inside the 24 FPS frame budget (41.7 ms), and the first frames within 0.25 RMS-relative of the
    bf16 model's hidden states (random-init weights: 36 layers amplify the 4 % operand noise of e4m3)."""
    import gc
    import time
    from infinitevl_amd.harness import GraphedDecode, GraphedStep, InfiniteVLTextConfig, InfiniteVLTextStack
    cfg = InfiniteVLTextConfig(sliding_window=4096)
    with torch.device(DEV):
        torch.set_default_dtype(torch.bfloat16)
        try:
            model = InfiniteVLTextStack(cfg)
        finally:
            torch.set_default_dtype(torch.float32)
    model = model.to(torch.bfloat16).eval()
    model.init_weights_(seed=0).fuse_()
    T = 256
    g_ = torch.Generator(device=DEV).manual_seed(11)
    frames = [(torch.randn(1, T, cfg.hidden_size, device=DEV, generator=g_) * 0.02).to(torch.bfloat16) for _ in range(3)]
    with torch.no_grad():
        # bf16 reference for the first frames (eager)
        cb = model.allocate_inference_cache(1)
        ref = []
        for i in range(3):
            pid = torch.arange(i * T, (i + 1) * T, device=DEV)[None, None, :].expand(3, 1, T)
            ref.append(model(inputs_embeds=frames[i], position_ids=pid, past_key_values=cb)[0].float().clone())
        del cb
        model.set_mma_dtype("fp8_e4m3")
        cache = model.allocate_inference_cache(1)
        gs = GraphedStep(model, cache, 1, T, logits_to_keep=1)
        for i in range(3):
            h, _ = gs.step(frames[i])
            e = parity.rms_rel(ref[i], h.float())
            assert 1e-3 < e < 0.25, (i, e)
        for i in range(3, 20):
            gs.step(frames[i % 3])
        torch.cuda.synchronize()
        gc.collect()
        base = torch.cuda.memory_allocated()
        dec = None
        t0 = time.perf_counter()
        for i in range(20, 4096):
            gs.step(frames[i % 3])
            if i % 512 == 0:                         # answer a question mid-stream: decode on a clone (fp8 SWA decode step)
                torch.cuda.synchronize()
                branch = cache.clone()
                dec = GraphedDecode(model, branch, 1)
                dec.token.copy_(gs.logits[:, -1].argmax(-1, keepdim=True))
                for _ in range(8):
                    dec.step()
                torch.cuda.synchronize()
                assert torch.isfinite(dec.logits.float()).all()
                del dec, branch
                gc.collect()
                torch.cuda.empty_cache()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        gc.collect()
        assert torch.cuda.memory_allocated() <= base, (torch.cuda.memory_allocated(), base)
        assert cache.get_seq_length() == 4096 * 256 == 1048576
        assert int(cache.layers[0]._pos_dev.item()) == 1048576
        assert torch.isfinite(gs.hidden.float()).all() and torch.isfinite(gs.logits.float()).all()
        assert wall / (4096 - 20) < 41.7e-3, wall
    model.set_mma_dtype(None)
    del model, gs, cache
    gc.collect()
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------
# boundary behaviour: foreign cache protocol, batches, error paths, weight fusion round trip
# ---------------------------------------------------------------------------------------------
class _CatCache:
    """A cache that follows the REFERENCE protocol (std:126-173): update() returns cat(cached, new) in
    [B,H,S,d] and keeps the last W-1 tokens.  Exercises the operator-level drop-in path of
    InfiniteVLSelfAttention (ALL_ATTENTION_FUNCTIONS-style call with concatenated K/V)."""

    def __init__(self, n_layers, window):
        self.layers = [object() for _ in range(n_layers)]
        self.k = self.v = None
        self.cap = window - 1

    def update(self, layer_idx, key_states, value_states, conv_state, recurrent_state, cache_kwargs):
        fk = key_states if self.k is None else torch.cat([self.k, key_states], dim=2)
        fv = value_states if self.v is None else torch.cat([self.v, value_states], dim=2)
        self.k, self.v = fk[:, :, -self.cap:].contiguous(), fv[:, :, -self.cap:].contiguous()
        return fk, fv


@pytest.mark.parametrize("fuse", [False, True])
def test_swa_module_with_reference_style_cat_cache_equals_ring_cache(fuse):
    stack, hc, _, _ = _small_stack(window=96, fuse=fuse)
    attn = stack.layers[0].self_attn
    rot = stack.rotary_emb
    ring = stack.allocate_inference_cache(2)
    cat = _CatCache(len(stack.layers), 96)
    pos = 0
    with torch.no_grad():
        for T in (50, 70, 1, 1, 33):
            x = bf(torch.randn(2, T, hc.hidden_size) * 0.5).to(DEV)
            pid = torch.arange(pos, pos + T, device=DEV)[None, None, :].expand(3, 2, T)
            pe = rot(x, pid)
            o1, _ = attn(x, position_ids=pid, past_key_values=ring, position_embeddings=pe)
            o2, _ = attn(x, position_ids=pid, past_key_values=cat, position_embeddings=pe)
            assert rms_rel(o2.float().cpu(), o1.float().cpu()) < 4e-3, T
            pos += T
    assert torch.equal(ring.layers[0].keys, cat.k)          # same tokens kept, same order, bit for bit


class _RefLinearCache:
    """A linear-attention cache that follows the REFERENCE protocol (std:286-340) without being our class: the first call
    returns ((None, None, None), None) whatever it holds (`start`, std:298-300); "get" hands out the pre-allocated
    cache-dtype tensors; "set" `.copy_`-downcasts what the operators RETURNED (new conv-state tensors, an fp32
    `final_state`) into them (std:314-335) and advances `seq_len`.  Drives the `native=False` branch of
    modules.GatedDeltaNet.forward: operators return fresh tensors, the cache does the bf16 rounding."""

    class _Layer:
        def __init__(self, B, H, K, V, Dq, Dk, Dv, W, dtype):
            mk = lambda *s: torch.full(s, float("nan"), dtype=dtype, device=DEV)  # noqa: E731   (torch.empty: std:277-284)
            self.conv_state_q, self.conv_state_k, self.conv_state_v = mk(B, Dq, W), mk(B, Dk, W), mk(B, Dv, W)
            self.recurrent_state = mk(B, H, K, V)
            self.start, self.seq_len = False, 0

    def __init__(self, n_layers, gdn, B, dtype=torch.bfloat16):
        H, K, V = gdn.num_heads, gdn.head_k_dim, gdn.head_v_dim
        self.layers = [self._Layer(B, H, K, V, H * K, gdn.key_dim, gdn.value_dim, gdn.conv_size, dtype) for _ in range(n_layers)]
        self.seen_fp32_state = False

    def update(self, layer_idx, key_states=None, value_states=None, conv_state=None, recurrent_state=None, cache_kwargs=None):
        L = self.layers[layer_idx]
        op = (cache_kwargs or {}).get("op", "get" if (conv_state is None and recurrent_state is None) else "set")
        if L.start is False:
            L.start = True
            return (None, None, None), None
        if op == "get":
            return (L.conv_state_q, L.conv_state_k, L.conv_state_v), L.recurrent_state
        for dst, src in zip((L.conv_state_q, L.conv_state_k, L.conv_state_v), conv_state):
            assert tuple(src.shape) == tuple(dst.shape)
            dst.copy_(src)
        assert tuple(recurrent_state.shape) == tuple(L.recurrent_state.shape)
        self.seen_fp32_state |= recurrent_state.dtype == torch.float32           # chunk_delta_h.py:291: fp32 final_state
        L.recurrent_state.copy_(recurrent_state)                                  # std:335: rounded to the cache dtype
        L.seq_len += int(cache_kwargs.get("delta_len", 0))
        return (L.conv_state_q, L.conv_state_k, L.conv_state_v), L.recurrent_state


def test_gdn_module_with_reference_protocol_linear_cache_equals_native_cache():
    """GatedDeltaNet (un-fused weights) against a foreign cache that speaks the reference's get / set protocol, beside our
    own in-place cache: chunk calls, a 64 / 65-token mode boundary (std:1230), recurrent calls and single tokens.  The
    operators return an fp32 final state and the cache rounds it with `.copy_` (std:325-335) where the native path has
    the kernel write bf16 in place - the same rounding point, so states and outputs agree to bf16 rounding of identical
    fp32 values (bit-equal states; outputs equal up to the kernels' own run-to-run determinism: bit-equal)."""
    stack, hc, _, _ = _small_stack(window=96, fuse=False)
    gdn = stack.layers[1].self_attn
    native = stack.allocate_inference_cache(2)
    ref = _RefLinearCache(len(stack.layers), gdn, 2)
    with torch.no_grad():
        for T in (130, 65, 64, 1, 1, 200, 7):
            x = bf(torch.randn(2, T, hc.hidden_size) * 0.5).to(DEV)
            o1, _ = gdn(x, past_key_values=native)
            o2, _ = gdn(x, past_key_values=ref)
            assert rms_rel(o2.float().cpu(), o1.float().cpu()) < 1e-6, T
            L1, L2 = native.layers[1], ref.layers[1]
            assert torch.equal(L1.recurrent_state, L2.recurrent_state), T
            for a, b in ((L1.conv_state_q, L2.conv_state_q), (L1.conv_state_k, L2.conv_state_k), (L1.conv_state_v, L2.conv_state_v)):
                assert torch.equal(a, b), T
    assert ref.seen_fp32_state and ref.layers[1].seq_len == native.layers[1].seq_len == 468


def test_ops_exports_every_name_the_reference_imports_from_fla():
    """std:52-54 imports eight names from fla; infinitevl_amd.ops provides all of them."""
    from infinitevl_amd import ops
    for name in ("get_unpad_data", "index_first_axis", "pad_input", "FusedRMSNormGated", "RMSNorm", "ShortConvolution",
                 "chunk_gated_delta_rule", "fused_recurrent_gated_delta_rule"):
        assert hasattr(ops, name), name
    # RMSNorm = the use_gate=False output norm (std:1213 / 1341), on the HIP norm kernel
    x = bf(torch.randn(3, 70, 4, 256)).to(DEV)
    norm = ops.RMSNorm(256, eps=1e-5).to(DEV, torch.bfloat16)
    with torch.no_grad():
        norm.weight.copy_(bf(torch.randn(256)))
        y = norm(x).float().cpu()
    xf, wf = x.float().cpu(), norm.weight.float().cpu()
    want = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * wf
    assert rms_rel(y, want) < 4e-3
    # unpad helpers: round trip through a padding mask (flash-attn bert_padding semantics)
    mask = torch.tensor([[1, 1, 1, 0], [1, 1, 0, 0]], device=DEV)
    idx, cu, mx = ops.get_unpad_data(mask)
    assert idx.tolist() == [0, 1, 2, 4, 5] and cu.tolist() == [0, 3, 5] and mx == 3 and cu.dtype == torch.int32
    h = torch.arange(8 * 3, device=DEV, dtype=torch.float32).view(8, 3)
    back = ops.pad_input(ops.index_first_axis(h, idx), idx, 2, 4)
    assert torch.equal(back.view(8, 3)[idx], h[idx]) and float(back.view(8, 3)[3].abs().sum()) == 0.0


def test_gdn_module_without_output_gate():
    """use_gate=False (std:1209-1213, 1336-1341): no g_proj, plain RMSNorm on the mixer output."""
    import copy as _copy
    from infinitevl_amd.modules import GatedDeltaNet
    stack, hc, _, _ = _small_stack(window=96, fuse=False)
    cfg = _copy.copy(hc)
    cfg.use_gate = False
    torch.manual_seed(0)
    m = GatedDeltaNet(cfg, 1).to(DEV, torch.bfloat16).eval()
    assert not hasattr(m, "g_proj") and type(m.o_norm).__name__ == "RMSNorm"
    g = stack.layers[1].self_attn
    sd = {k: v for k, v in g.state_dict().items() if not k.startswith("g_proj")}
    m.load_state_dict(sd)
    x = bf(torch.randn(1, 130, hc.hidden_size) * 0.5).to(DEV)
    with torch.no_grad():
        o, _ = m(x)
        # the same mixer core with the gate: undo gate and norm by recomputing from the core output is not possible from
        # outside, so check the structure instead: finite, right shape, and equal to o_proj(rmsnorm(core)) rebuilt by hand
        from infinitevl_amd import ops
        q, _ = m.q_conv1d(m.q_proj(x)); k, _ = m.k_conv1d(m.k_proj(x)); v, _ = m.v_conv1d(m.v_proj(x))
        gg, beta = ops.gdn_gate(m.a_proj(x), m.b_proj(x), m.A_log, m.dt_bias)
        core, _ = ops.chunk_gated_delta_rule(q.view(1, 130, m.num_heads, -1), k.view(1, 130, m.num_heads, -1),
                                             v.view(1, 130, m.num_heads, -1), gg, beta, use_qk_l2norm_in_kernel=True)
        want = m.o_proj(m.o_norm(core).reshape(1, 130, -1))
    assert o.shape == x.shape and torch.equal(o, want)


def test_plain_rmsnorm_has_flas_single_rounding():
    """ops.RMSNorm at head_v_dim = 256 (the mixer's output norm when use_gate=False): fla's arithmetic -- x * rstd * w in fp32,
    rounded to bf16 once (fla:modules/layernorm.py:128-143) -- bit for bit against the same expression in torch fp32; the
    Qwen2RMSNorm form (x * rstd rounded first) differs from it in a few per cent of the elements."""
    from infinitevl_amd import ops
    g_ = torch.Generator(device=DEV).manual_seed(5)
    x = bf(torch.randn(3, 70, 4, 256, device=DEV, generator=g_) * 3.0)
    m = ops.RMSNorm(256, eps=1e-6, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        m.weight.copy_(bf(torch.randn(256, device=DEV, generator=g_)))
        y = m(x)
    xf, wf = x.float(), m.weight.float()
    rstd = 1.0 / torch.sqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
    want = bf(xf * rstd * wf)
    mism = (y != want).float().mean().item()
    assert rms_rel(y, (xf * rstd * wf)) < 3e-3
    assert mism < 2e-3, mism                      # (the kernel's statistic sums in another order than torch's mean: rare 1-ulp flips)
    qwen = bf(wf * bf(xf * rstd).float())
    assert (qwen != want).float().mean().item() > 5 * max(mism, 1e-3)


def test_cache_error_behaviour_on_gpu():
    from infinitevl_amd.cache import StaticCachePrealloc
    stack, hc, _, _ = _small_stack(window=96)
    cache = StaticCachePrealloc(config=hc, batch_size=2, device=DEV, dtype=torch.bfloat16)
    k = torch.zeros(1, 5, 1, 128, dtype=torch.bfloat16, device=DEV)
    q = torch.zeros(1, 5, 2, 128, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ValueError, match="pre-allocated batch_size=2"):          # std:137-138
        cache.layers[0].attend(q, k, k, 1.0, 96)
    k2 = torch.zeros(2, 5, 2, 128, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ValueError, match="head dim mismatch"):                   # std:139-140
        cache.layers[0].attend(torch.zeros(2, 5, 2, 128, dtype=torch.bfloat16, device=DEV), k2, k2, 1.0, 96)
    with pytest.raises(ValueError, match="pre-allocated batch_size=2"):
        cache.layers[0].update(k.transpose(1, 2), k.transpose(1, 2))


def test_fuse_keeps_state_dict_and_reloads():
    """fuse_() must not change parameter names/shapes/values, and loading a checkpoint AFTER fusing must
    update the fused tensors (parameters are views of them)."""
    from oracle import model as omodel
    stack, hc, oc, params = _small_stack(window=96, fuse=False)
    before = {k: v.clone() for k, v in stack.state_dict().items()}
    stack.fuse_()
    after = stack.state_dict()
    assert list(before) == list(after)
    assert all(torch.equal(before[k], after[k]) for k in before)
    new = parity.bf16_params(omodel.random_params(oc, seed=99, vocab=hc.vocab_size))
    stack.load_state_dict({k: v.to(torch.bfloat16) for k, v in new.items()}, strict=False)
    gdn = stack.layers[1].self_attn
    assert gdn.q_proj.weight.data_ptr() == gdn._fused_w.data_ptr()                # still views
    assert torch.equal(gdn._fused_w[gdn._fused_cols[5]:gdn._fused_cols[5] + 2].float().cpu(),
                       new["layers.1.self_attn.b_proj.weight"])
    x = bf(torch.randn(1, 70, hc.hidden_size) * 0.5)
    pid = torch.arange(70)[None, None, :].expand(3, 1, 70).contiguous()
    with torch.no_grad():
        h, _ = stack(inputs_embeds=x.to(DEV), position_ids=pid.to(DEV), past_key_values=stack.allocate_inference_cache(1))
    h_ref = omodel.text_stack(new, x.float(), pid, oc, omodel.new_cache(oc, torch.bfloat16), torch.bfloat16, torch.bfloat16)
    assert rms_rel(h_ref, h.float().cpu()) < 1.5e-2


def test_fp32_cache_state_option():
    """The build may keep the carried GDN state in fp32 (SURVEY.md Q5 option): a float32 cache works through the
    same modules (conv states converted, recurrent state written in place as fp32)."""
    stack, hc, _, _ = _small_stack(window=96, fuse=True)
    c32 = stack.allocate_inference_cache(1, dtype=torch.float32)
    c16 = stack.allocate_inference_cache(1)
    with torch.no_grad():
        pos = 0
        for T in (130, 70, 1):
            x = bf(torch.randn(1, T, hc.hidden_size) * 0.5).to(DEV)
            pid = torch.arange(pos, pos + T, device=DEV)[None, None, :].expand(3, 1, T)
            # SWA ring stays bf16-only: give the fp32 cache a bf16 SWA layer
            c32.layers[0] = c16.layers[0].clone() if pos == 0 else c32.layers[0]
            h32, _ = stack(inputs_embeds=x, position_ids=pid, past_key_values=c32)
            h16, _ = stack(inputs_embeds=x, position_ids=pid, past_key_values=c16)
            assert rms_rel(h32.float().cpu(), h16.float().cpu()) < 2e-2
            pos += T
    assert c32.layers[1].recurrent_state.dtype == torch.float32
    assert rms_rel(c32.layers[1].recurrent_state.cpu(), c16.layers[1].recurrent_state.float().cpu()) < 2e-2


def test_batched_streams_are_independent():
    """Batch sharding premise: sequences in a batch do not interact -- B=3 run together ==
    each sequence run alone, bit for bit (GDN chunk + recurrent, SWA ring, conv carry-in)."""
    stack, hc, _, _ = _small_stack(window=96, fuse=True)
    xs = [bf(torch.randn(3, T, hc.hidden_size) * 0.5).to(DEV) for T in (130, 70, 1, 1)]
    with torch.no_grad():
        cb = stack.allocate_inference_cache(3)
        singles = [stack.allocate_inference_cache(1) for _ in range(3)]
        pos = 0
        for x in xs:
            T = x.shape[1]
            pid = torch.arange(pos, pos + T, device=DEV)[None, None, :]
            hb, _ = stack(inputs_embeds=x, position_ids=pid.expand(3, 3, T), past_key_values=cb)
            for i in range(3):
                hi, _ = stack(inputs_embeds=x[i:i + 1], position_ids=pid.expand(3, 1, T), past_key_values=singles[i])
                assert rms_rel(hi.float().cpu(), hb[i:i + 1].float().cpu()) < 4e-3, (T, i)    # GEMM M differs -> fp32 order
            pos += T


# ---------------------------------------------------------------------------------------------
# decode-step projections (single-token weight stream)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,with_bias", [(1, 2048, 2048, False), (1, 22016, 2048, False), (1, 2048, 11008, False),
                                              (2, 12320, 2048, False), (4, 2560, 2048, True), (1, 151936, 2048, False),
                                              (3, 100, 136, True), (1, 7, 8, False)])
def test_linear_small_m_vs_fp32(M, N, K, with_bias):
    """ivl_linear_small_m_fwd against a plain fp32 torch reference (a floating-point kernel: tolerance = half a
    bf16 ulp of the output plus fp32 summation-order noise)."""
    from infinitevl_amd import ops
    torch.manual_seed(M * 1000 + N + K)
    x = bf(torch.randn(M, 1, K)).to(DEV)
    w = bf(torch.randn(N, K) * K ** -0.5).to(DEV)
    b = bf(torch.randn(N)).to(DEV) if with_bias else None
    y = ops.linear(x, w, b)
    assert y.shape == (M, 1, N) and y.dtype == torch.bfloat16
    ref = x.double().cpu() @ w.double().cpu().T + (b.double().cpu() if with_bias else 0.0)
    err = (y.double().cpu() - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-5                 # bf16: 8 significant bits -> half ulp = 2^-9 relative
    assert bool((err <= tol).all()), float((err / (ref.abs() + 1e-3)).max())


@pytest.mark.parametrize("M,I,K", [(1, 11008, 2048), (2, 256, 512), (4, 5000, 136), (1, 3, 8)])
def test_linear_swiglu_equals_linear_plus_gate_kernel(M, I, K):
    """The gate applied in the epilogue of the weight-stream kernel == ivl_linear_small_m_fwd on the fused gate|up
    weight followed by ivl_silu_mul_fwd, bit for bit."""
    from infinitevl_amd import ops
    torch.manual_seed(M + I + K)
    x = bf(torch.randn(M, 1, K)).to(DEV)
    w = bf(torch.randn(2 * I, K) * K ** -0.5).to(DEV)
    if I % 8 == 0:
        assert torch.equal(ops.linear_swiglu(x, w), ops.silu_mul(ops.linear(x, w)))
    else:       # silu_mul wants I % 8 == 0: compare with the same arithmetic in torch
        lin = ops.linear(x, w).float()
        g, u = lin[..., :I], lin[..., I:]
        ref = (bf((g * torch.sigmoid(g)).cpu()).to(DEV) * u).to(torch.bfloat16)
        assert rms_rel(ref.float().cpu(), ops.linear_swiglu(x, w).float().cpu()) < 4e-3


def test_linear_small_m_error_behaviour():
    from infinitevl_amd import _lib
    lib = _lib.load()
    x = torch.zeros(8, 64, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(16, 64, dtype=torch.bfloat16, device=DEV)
    y = torch.zeros(8, 16, dtype=torch.bfloat16, device=DEV)
    assert lib.ivl_linear_small_m_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 5, 16, 64, None) == _lib.IVL_ERR_UNSUPPORTED
    assert lib.ivl_linear_small_m_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 1, 16, 60, None) == _lib.IVL_ERR_INVALID_ARG
    assert lib.ivl_linear_small_m_fwd(None, w.data_ptr(), None, y.data_ptr(), 1, 16, 64, None) == _lib.IVL_ERR_INVALID_ARG
    # norm + projection: the norm weight is mandatory, a residual needs somewhere to put the new residual stream, K <= 4096
    nw = torch.ones(64, dtype=torch.bfloat16, device=DEV)
    xp, wp, yp, nwp = x.data_ptr(), w.data_ptr(), y.data_ptr(), nw.data_ptr()
    assert lib.ivl_norm_linear_small_m_fwd(xp, None, None, 1e-6, None, wp, None, yp, 1, 16, 64, 0, None) == _lib.IVL_ERR_INVALID_ARG
    assert lib.ivl_norm_linear_small_m_fwd(xp, xp, nwp, 1e-6, None, wp, None, yp, 1, 16, 64, 0, None) == _lib.IVL_ERR_INVALID_ARG
    assert lib.ivl_norm_linear_small_m_fwd(xp, None, nwp, 1e-6, None, wp, None, yp, 1, 16, 8192, 0, None) == _lib.IVL_ERR_UNSUPPORTED
    assert lib.ivl_norm_linear_small_m_fwd(xp, None, nwp, 1e-6, None, wp, None, yp, 5, 16, 64, 0, None) == _lib.IVL_ERR_UNSUPPORTED


def test_gdn_chunk_fused_rejects_a_misaligned_sync_area():
    """ivl_gdn_chunk_fused_fwd: the optional sync area must be 16-byte aligned (IVL_ERR_INVALID_ARG -> ValueError)."""
    from infinitevl_amd import ops
    H, K, V, T = 16, 128, 256, 64
    Dq, Dk, Dv = H * K, H * K, H * V
    cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
    proj = torch.zeros(1, T, cols[4] + H, dtype=torch.bfloat16, device=DEV)
    cw = [torch.zeros(D_, 1, 4, dtype=torch.bfloat16, device=DEV) for D_ in (Dq, Dk, Dv)]
    A32, dt32 = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    good = ops._gdn_sync_area(torch.device(DEV))
    bad = torch.zeros(good.numel() + 16, dtype=torch.uint8, device=DEV)[4:]
    with ops.gdn_sync_scope(bad), pytest.raises(ValueError):
        ops.gdn_chunk_fused(proj, cols, cw, [None] * 3, [None] * 3, A32, dt32, H, K, V)


def test_decode_step_uses_weight_stream_and_matches_gemm_path():
    """One decode token through the fused 4-layer stack: the small-M projections (ivl_linear_small_m_fwd) against the
    same step with the projections forced through the library GEMM (x padded to 5 rows)."""
    from infinitevl_amd import ops
    from infinitevl_amd.harness import InfiniteVLTextStack
    hc, _ = parity.small_configs(window=96)
    torch.manual_seed(2)
    stack = InfiniteVLTextStack(hc).to(device=DEV, dtype=torch.bfloat16).eval()
    stack.init_weights_(seed=3)
    stack.fuse_()
    caches = [stack.allocate_inference_cache(1) for _ in range(2)]
    x0 = bf(torch.randn(1, 40, hc.hidden_size) * 0.5).to(DEV)
    x1 = bf(torch.randn(1, 1, hc.hidden_size) * 0.5).to(DEV)
    pid0 = torch.arange(40, device=DEV)[None, None].expand(3, 1, 40).contiguous()
    pid1 = torch.full((3, 1, 1), 40, device=DEV)
    outs = []
    real_linear = ops.linear
    try:
        for forced in (False, True):
            if forced:
                ops.linear = lambda x, w, b=None: torch.nn.functional.linear(          # noqa: E731
                    x.materialize() if isinstance(x, ops.PreNorm) else x, w, b)
            with torch.no_grad():
                stack(inputs_embeds=x0, position_ids=pid0, past_key_values=caches[forced], logits_to_keep=1)
                h, lg = stack(inputs_embeds=x1, position_ids=pid1, past_key_values=caches[forced], logits_to_keep=1)
            outs.append((h.float().cpu(), lg.float().cpu()))
    finally:
        ops.linear = real_linear
    assert rms_rel(outs[1][0], outs[0][0]) < 1e-2 and rms_rel(outs[1][1], outs[0][1]) < 1e-2


@pytest.mark.parametrize("B,H,state_dtype", [(1, 16, torch.bfloat16), (2, 3, torch.float32), (3, 2, torch.bfloat16)])
def test_gdn_decode_step_equals_three_kernel_sequence(B, H, state_dtype):
    """ivl_gdn_decode_step_fwd == ivl_gdn_prologue_fwd -> ivl_gdn_recurrent_fwd -> ivl_rmsnorm_swish_gate_strided_fwd
    on the same fused projection row: conv states bit for bit (pure data movement), outputs / recurrent state to
    fp32 summation order (the reductions are partitioned differently)."""
    from infinitevl_amd import ops
    K, V = 128, 256
    Dq, Dv = H * K, H * V
    cols = (0, Dq, 2 * Dq, 2 * Dq + Dv, 2 * Dq + 2 * Dv, 2 * Dq + 2 * Dv + H)      # q k v g a b
    ld = cols[5] + H
    ld += (-ld) % 8
    torch.manual_seed(B * 100 + H)
    state0 = bf(torch.randn(B, H, K, V)).to(DEV, state_dtype)
    conv0 = [bf(torch.randn(B, D, 4)).to(DEV) for D in (Dq, Dq, Dv)]
    cw = [bf(torch.randn(D, 1, 4) * 0.5).to(DEV) for D in (Dq, Dq, Dv)]
    A32, dt32 = torch.randn(H).to(DEV), torch.randn(H).to(DEV)
    wn = bf(torch.randn(V)).to(DEV)
    for step in range(3):
        proj = bf(torch.randn(B, 1, ld)).to(DEV)
        if step == 0:
            st_a, st_b = state0.clone(), state0.clone()
            ca, cb = [c.clone() for c in conv0], [c.clone() for c in conv0]
        # (a) three kernels
        q, k, v, g, beta = ops.gdn_prologue(proj, (cols[0], cols[1], cols[2], cols[4], cols[5]), cw, ca, ca, A32, dt32,
                                            H, Dq, Dq, Dv)
        o, _ = ops.fused_recurrent_gated_delta_rule(q.view(B, 1, H, K), k.view(B, 1, H, K), v.view(B, 1, H, V), g, beta,
                                                    initial_state=st_a, use_qk_l2norm_in_kernel=True, final_state_out=st_a)
        y_a = ops.rmsnorm_swish_gate_strided(o, proj[..., cols[3]:], ld, wn, 1e-5).reshape(B, 1, Dv)
        # (b) one launch
        y_b = ops.gdn_decode_step(proj, cols, cw, cb, A32, dt32, wn, 1e-5, st_b, H, K, V, K ** -0.5)
        for x_a, x_b in zip(ca, cb):
            assert torch.equal(x_a, x_b), step
        assert rms_rel(y_a.float().cpu(), y_b.float().cpu()) < 3e-3, step
        assert rms_rel(st_a.float().cpu(), st_b.float().cpu()) < (3e-3 if state_dtype == torch.bfloat16 else 1e-5), step
    assert torch.isfinite(y_b.float()).all()


def test_carried_state_import_equals_continuing_on_the_same_cache():
    """The hand-off contract of SURVEY.md 8f-4 inside one process: copy carried_tensors() of every layer into a fresh
    cache + import_carried(), continue there; equals continuing on the original cache bit for bit."""
    from infinitevl_amd.harness import InfiniteVLTextStack
    hc, _ = parity.small_configs(window=96)
    stack = InfiniteVLTextStack(hc).to(device=DEV, dtype=torch.bfloat16).eval()
    stack.init_weights_(seed=7)
    stack.fuse_()
    torch.manual_seed(1)
    xs = bf(torch.randn(1, 200, hc.hidden_size) * 0.5).to(DEV)
    a, b = stack.allocate_inference_cache(1), stack.allocate_inference_cache(1)
    with torch.no_grad():
        stack(inputs_embeds=xs[:, :128], past_key_values=a, logits_to_keep=0)
        for la, lb in zip(a.layers, b.layers):
            for ta, tb in zip(la.carried_tensors(), lb.carried_tensors()):
                tb.copy_(ta)
            lb.import_carried(128)
        pos = torch.arange(128, 200, device=DEV)[None, None].expand(3, 1, 72).contiguous()
        h_b, _ = stack(inputs_embeds=xs[:, 128:], position_ids=pos, past_key_values=b, logits_to_keep=0)
        h_a, _ = stack(inputs_embeds=xs[:, 128:], past_key_values=a, logits_to_keep=0)
    assert torch.equal(h_a, h_b)
    for la, lb in zip(a.layers, b.layers):
        for ta, tb in zip(la.carried_tensors(), lb.carried_tensors()):
            assert torch.equal(ta, tb)
    assert a.get_seq_length() == b.get_seq_length() == 200


def test_sequence_parallel_prefill_two_ranks_bit_exact():
    """one sequence cut in two segments, two processes, per-layer state hand-off (tools/sp_check.py);
    the result equals the single-process two-call sequence bit for bit (hidden, cache, following decode steps)."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "sp.pt")
        r = subprocess.run([_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "sp_check.py"),
                            "--dump", dump], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "SP_CHECK PASS" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        z = torch.load(dump)
    # ... and against the ORACLE: the whole sequence in one CPU pass of the restated model (same weights, the reference's bf16
    # rounding points), compared on the last rank's segment -- the hand-off path is checked against the reference's algorithm,
    # not only against the HIP path's own single-process sequence
    from oracle import model as omodel
    hc, oc = parity.small_configs(z["window"], n_layers=z["layers"], heads=z["heads"])
    params = {k: v for k, v in z["state_dict"].items() if "inv_freq" not in k}
    total = z["xs"].shape[1]
    pid = torch.arange(total)[None, None, :].expand(3, 1, total)
    h_ref = omodel.text_stack(params, z["xs"], pid, oc, omodel.new_cache(oc, cache_dtype=torch.bfloat16), act_dtype=torch.bfloat16)
    h_ref = h_ref[0] if isinstance(h_ref, tuple) else h_ref
    err = rms_rel(h_ref[:, z["first"]:z["last"]], z["h_last_segment"])
    assert err < 3e-2, err


def test_bench_two_ranks_reports_what_the_collective_ran_on():
    """bench.py --gpus 2 (self-spawned under torch.distributed.run) with the gloo debug backend, both ranks on this GPU: the
    JSON line carries a `dist` object (backend, world, one record per rank with device index and PCI bus id, the gathered
    logits' shape, the all-gather time) - the fields that make a driver-run multi-GPU line checkable from its output.  Under
    RCCL the same code asserts that the ranks sit on distinct devices."""
    import json as _json
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IVL_DIST_BACKEND="gloo", IVL_NO_TUNABLEOP="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--layers", "4", "--context", "8192", "--decode-steps", "2",
                        "--no-fp8", "--no-kernel-timing", "--sp-tokens", "1024"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = _json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 and out["scaling"] == "weak"
    d = out["dist"]
    assert d["backend"] == "gloo" and d["world"] == 2 and len(d["devices"]) == 2
    assert [x["rank"] for x in d["devices"]] == [0, 1] and all(x["pci_bus_id"] for x in d["devices"])
    assert d["gathered_logits_shape"] == [2, 151936] and d["allgather_ms"] > 0 and d["logits_rows_differ_across_ranks"]
    assert d["distinct_devices"] == 1          # two gloo ranks stacked on one GPU; RCCL runs assert == world
    # the sequence-parallel prefill leg rides on the same process group: one 2 x 1024-token sequence, the last
    # rank's last-token logits equal its own single-rank run of the same two calls bit for bit; under gloo the hand-off is
    # host-staged and the record says so
    sp = d["sp"]
    assert sp["tokens"] == 2048 and sp["backend"] == "gloo" and sp["device_p2p"] is False and "host-staged" in sp["transport"]
    assert sp["last_token_logits_equal_single_rank_run"] is True and sp["max_abs_diff"] == 0.0 and sp["logits_finite"]
    assert sp["ms"] > 0 and sp["tok_s"] > 0 and sp["single_rank_ms_same_calls"] > 0 and sp["failed"] is False
    assert out["logits_finite"] and out["value"] > 0
    # at N > 1 the line is COMPLETE -- the configs[3] leg runs on EVERY rank (one sequence per GPU = "batch
    # sharded"), with one record per rank, the slowest rank's barrier-to-barrier time, the aggregate over all ranks and the one
    # logits gather; the configs[1] leg likewise; cpu_baseline stays on rank 0
    c3 = out["cfg3_512k_prefill"]
    assert c3.get("failed") is not True, c3
    assert c3["n_gpus"] == 2 and c3["sequences"] == 2 and len(c3["per_rank_ms"]) == 2 and all(v > 0 for v in c3["per_rank_ms"])
    assert c3["tokens_timed_all_ranks"] == 2 * c3["calls_timed"] * 4096 and abs(c3["ms_per_call"] - max(c3["per_rank_ms"])) < 1e-3
    assert abs(c3["aggregate_tok_s"] - c3["tokens_timed_all_ranks"] / (c3["region_ms_max_over_ranks"] * 1e-3)) < 1e-3 * c3["aggregate_tok_s"]
    assert c3["region_ms_max_over_ranks"] >= c3["calls_timed"] * min(c3["per_rank_ms"]) * 0.99
    assert c3["gathered_logits_shape"] == [2, 151936] and c3["gather_ms"] > 0 and c3["logits_finite"] and c3["constant_memory"]
    c1 = out["cfg1_4k_prefill_decode"]
    assert c1["n_gpus"] == 2 and c1["prefill_tok_s"] > 0 and c1["decode_tok_s"] > 0 and c1["logits_finite"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "oracle/model.py" in cb["sample"]


def test_full_attention_layer_with_dynamic_cache_equals_one_causal_call():
    """A "full_attention" layer type (strm:548-550 fallback): the module goes through the growing DynamicLayer cache
    and plain causal attention; two chunks fed through the cache == one call over the whole sequence."""
    from infinitevl_amd.cache import DynamicLayer, StaticCachePrealloc
    from infinitevl_amd.harness import InfiniteVLTextConfig
    from infinitevl_amd.modules import InfiniteVLRotaryEmbedding, InfiniteVLSelfAttention
    cfg = InfiniteVLTextConfig(vocab_size=64, hidden_size=256, intermediate_size=512, num_hidden_layers=1,
                               num_attention_heads=2, num_key_value_heads=1, head_dim=128, sliding_window=16,
                               layer_types=["full_attention"])
    torch.manual_seed(4)
    attn = InfiniteVLSelfAttention(cfg, 0).to(DEV, torch.bfloat16).eval()
    rope = InfiniteVLRotaryEmbedding(cfg).to(DEV)
    assert attn.sliding_window is None
    x = bf(torch.randn(1, 150, 256) * 0.5).to(DEV)
    pos = torch.arange(150, device=DEV)[None, None].expand(3, 1, 150).contiguous()
    with torch.no_grad():
        full, _ = attn(x, position_embeddings=rope(x, pos))
        cache = StaticCachePrealloc(config=cfg, batch_size=1, device=DEV, dtype=torch.bfloat16)
        assert isinstance(cache.layers[0], DynamicLayer)
        a, _ = attn(x[:, :90], position_embeddings=rope(x[:, :90], pos[:, :, :90]), past_key_values=cache)
        b, _ = attn(x[:, 90:], position_embeddings=rope(x[:, 90:], pos[:, :, 90:]), past_key_values=cache)
    assert cache.get_seq_length() == 150
    got = torch.cat([a, b], 1)
    assert rms_rel(full.float().cpu(), got.float().cpu()) < 4e-3


@pytest.mark.parametrize("B,T,hist,st_dtype", [(1, 256, True, torch.bfloat16), (2, 130, False, torch.float32), (1, 65, True, torch.float32),
                                              (1, 3, True, torch.bfloat16), (1, 1000, True, torch.bfloat16),
                                              # 64-column scan workgroups (B H > 32: the V waves load their rows themselves)
                                              (3, 200, True, torch.bfloat16), (4, 70, False, torch.float32),
                                              # two workspace segments (68 chunks): value rows / history of a segment that does not start at 0
                                              (1, 4300, True, torch.bfloat16)])
def test_gdn_chunk_with_fused_front_end_is_bit_identical(B, T, hist, st_dtype):
    """the pre-pass that applies the three short convs (+SiLU, carry-in / carry-out) and the gate math
    itself (ivl_gdn_chunk_fused_fwd) must equal ivl_gdn_prologue_fwd followed by ivl_gdn_chunk_fwd bit for bit: outputs,
    final recurrent state and the three new conv states, with and without history, ragged T, T below the conv width."""
    from infinitevl_amd import ops
    H, K, V = 16, 128, 256
    Dq, Dk, Dv = H * K, H * K, H * V
    g_ = torch.Generator(device=DEV).manual_seed(B * 1000 + T)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)          # q | k | v | gate (unused here) | a | b
    ld = cols[4] + H
    proj = rn(B, T, ld)
    cw = [rn(D_, 1, 4) * 0.5 for D_ in (Dq, Dk, Dv)]
    A32, dt32 = torch.randn(H, device=DEV, generator=g_), torch.randn(H, device=DEV, generator=g_)
    cs = [rn(B, D_, 4) if hist else None for D_ in (Dq, Dk, Dv)]
    h0 = (torch.randn(B, H, K, V, device=DEV, generator=g_) * 0.1).to(st_dtype) if hist else None
    # unfused
    so1 = [torch.zeros(B, D_, 4, dtype=torch.bfloat16, device=DEV) for D_ in (Dq, Dk, Dv)]
    q, k, v, g, beta = ops.gdn_prologue(proj, cols, cw, cs, so1, A32, dt32, H, Dq, Dk, Dv)
    ht1 = torch.zeros(B, H, K, V, dtype=st_dtype, device=DEV)
    o1, _ = ops.chunk_gated_delta_rule(q.view(B, T, H, K), k.view(B, T, H, K), v.view(B, T, H, V), g, beta, initial_state=h0,
                                       use_qk_l2norm_in_kernel=True, final_state_out=ht1)
    # fused (conv states updated IN PLACE when there is history: out aliases in)
    so2 = [c.clone() for c in cs] if hist else [torch.zeros(B, D_, 4, dtype=torch.bfloat16, device=DEV) for D_ in (Dq, Dk, Dv)]
    ht2 = torch.zeros(B, H, K, V, dtype=st_dtype, device=DEV)
    o2 = ops.gdn_chunk_fused(proj, cols, cw, so2 if hist else cs, so2, A32, dt32, H, K, V, initial_state=h0, final_state_out=ht2)
    assert torch.equal(o1, o2)
    assert torch.equal(ht1, ht2)
    for a, b_ in zip(so1, so2):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("B,T,hist,mma,H", [(1, 256, True, None, 16), (1, 512, True, None, 16), (1, 64, False, None, 16), (1, 300, True, None, 16),
                                            (1, 7, True, None, 16), (1, 256, True, "fp8_e4m3", 16), (2, 128, True, None, 16),
                                            # long calls: persistent pre-pass workgroups beside the scan, records awaited chunk by chunk; two workspace segments
                                            (1, 1000, True, None, 16), (1, 4300, True, None, 16),
                                            # ... the same in e4m3 (the long-call launch holds TWO pre-pass bodies of 70 KB whatever the scan's
                                            # own LDS need is: the fp8 scan images are half the bf16 ones)
                                            (1, 1000, True, "fp8_e4m3", 16), (1, 4300, True, "fp8_e4m3", 16),
                                            # B*H not a multiple of 8: a head's pre-pass and scan workgroups sit on DIFFERENT dies (records cross L2s)
                                            (1, 256, True, None, 3), (3, 200, True, None, 5), (1, 1000, True, None, 6), (1, 8292, True, None, 1)])
def test_gdn_chunk_fused_single_launch_equals_two_launches(B, T, hist, mma, H):
    """ivl_gdn_chunk_fused_fwd with a sync area (pre-pass and scan workgroups of ONE launch, the scan side waiting on flags) must
    equal the two-launch form bit for bit -- outputs, final state, conv states -- call after call (the launch clears its own
    flags: the area is all-zero afterwards) and when replayed from a hipGraph."""
    from infinitevl_amd import ops
    K, V = 128, 256
    Dq, Dk, Dv = H * K, H * K, H * V
    g_ = torch.Generator(device=DEV).manual_seed(B * 77 + T)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
    ld = (cols[4] + H + 7) // 8 * 8                                      # rows are 16-byte aligned
    cw = [rn(D_, 1, 4) * 0.5 for D_ in (Dq, Dk, Dv)]
    A32, dt32 = torch.randn(H, device=DEV, generator=g_), torch.randn(H, device=DEV, generator=g_)

    def run(proj, cs, h0, single):
        so = [c.clone() for c in cs] if hist else [torch.zeros(B, D_, 4, dtype=torch.bfloat16, device=DEV) for D_ in (Dq, Dk, Dv)]
        ht = torch.zeros(B, H, K, V, dtype=torch.bfloat16, device=DEV)
        ops._GDN_SINGLE_LAUNCH = single
        try:
            o = ops.gdn_chunk_fused(proj, cols, cw, so if hist else [None] * 3, so, A32, dt32, H, K, V, initial_state=h0, final_state_out=ht,
                                    mma_dtype=mma)
        finally:
            ops._GDN_SINGLE_LAUNCH = True
        return o, ht, so

    area = ops._gdn_sync_area(torch.device(DEV))
    # the two-launch reference of every input set first, then the single-launch calls back to back: a record line left in
    # some L2 by an earlier launch would belong to different inputs
    sets, refs = [], []
    for it in range(10 if T <= 512 else 3):
        st = (rn(B, T, ld), [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)],
              bf(torch.randn(B, H, K, V, device=DEV, generator=g_) * 0.1) if hist else None)
        sets.append(st)
        refs.append(run(*st, False))
    for rep in range(2):
        outs = [run(*st, True) for st in sets]
        for it, ((o2, ht2, so2), (o1, ht1, so1)) in enumerate(zip(refs, outs)):
            assert torch.equal(o1, o2) and torch.equal(ht1, ht2), (rep, it)
            for a, b_ in zip(so1, so2):
                assert torch.equal(a, b_)
        assert int(area.view(torch.int32).abs().sum()) == 0, "the launch must leave its flags cleared"
    # hipGraph replay of the single-launch form: static inputs rewritten between replays
    proj = rn(B, T, ld)
    so = [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)]
    st = bf(torch.randn(B, H, K, V, device=DEV, generator=g_) * 0.1)
    o_static = torch.empty(B, T, H, V, dtype=torch.bfloat16, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gdn_chunk_fused(proj, cols, cw, [c.clone() for c in so], [c.clone() for c in so], A32, dt32, H, K, V, initial_state=st.clone(),
                            final_state_out=st.clone())
    torch.cuda.current_stream().wait_stream(side)
    ops.prepare_gdn_capture()          # a capture outside any gdn_sync_scope: the records workspace of the device's graph area must exist
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o_static.copy_(ops.gdn_chunk_fused(proj, cols, cw, so, so, A32, dt32, H, K, V, initial_state=st, final_state_out=st))
    for it in range(6):
        proj.copy_(rn(B, T, ld))
        so_ref = [c.clone() for c in so]
        st_ref = st.clone()
        ops._GDN_SINGLE_LAUNCH = False
        try:
            o_ref = ops.gdn_chunk_fused(proj, cols, cw, so_ref, so_ref, A32, dt32, H, K, V, initial_state=st_ref, final_state_out=st_ref)
        finally:
            ops._GDN_SINGLE_LAUNCH = True
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o_static, o_ref) and torch.equal(st, st_ref), it
        for a, b_ in zip(so, so_ref):
            assert torch.equal(a, b_)
    assert int(area.view(torch.int32).abs().sum()) == 0


def _gdn_fused_case(T, H=16, B=1, seed=0):
    """inputs of one fused GDN call + a runner (single-launch form or two launches)"""
    from infinitevl_amd import ops
    K, V = 128, 256
    Dq, Dk, Dv = H * K, H * K, H * V
    g_ = torch.Generator(device=DEV).manual_seed(9000 + seed + T)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
    cw = [rn(D_, 1, 4) * 0.5 for D_ in (Dq, Dk, Dv)]
    A32, dt32 = torch.randn(H, device=DEV, generator=g_), torch.randn(H, device=DEV, generator=g_)
    proj = rn(B, T, cols[4] + H)
    cs = [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)]
    h0 = bf(torch.randn(B, H, K, V, device=DEV, generator=g_) * 0.1)

    def run(single, ht_fill=None):
        so = [c.clone() for c in cs]
        ht = torch.zeros_like(h0) if ht_fill is None else torch.full_like(h0, ht_fill)
        ops._GDN_SINGLE_LAUNCH = single
        try:
            o = ops.gdn_chunk_fused(proj, cols, cw, so, so, A32, dt32, H, K, V, initial_state=h0, final_state_out=ht)
        finally:
            ops._GDN_SINGLE_LAUNCH = True
        return o, ht, so
    return run


def _gdn_kernel_names(fn):
    """names of the GDN chunk kernels `fn` launches (profiler activity records)"""
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return [e.name for e in prof.events() if "gdn_chunk" in e.name]


def test_gdn_single_launch_is_gated_on_real_occupancy():
    """The single-launch forms are taken only when the occupancy query says the grid can be resident at once
    (ivl_gdn_resident_blocks stands in for a smaller device): a too-small device takes the two-launch form, a device with room
    for the unsplit grid only takes the unsplit form -- always with the same bits."""
    from infinitevl_amd import ops
    run = _gdn_fused_case(256)
    ref = run(False)

    def same(out, r=ref):
        return torch.equal(out[0], r[0]) and torch.equal(out[1], r[1]) and all(torch.equal(a, b_) for a, b_ in zip(out[2], r[2]))
    env_override = os.environ.get("IVL_GDN_RESIDENT_BLOCKS", "").strip()       # what the process was started with is restored at the end
    restore = int(env_override) if env_override else -1
    try:
        assert ops.gdn_resident_blocks(-1) == 256, "a whole MI355X holds one single-launch workgroup per CU"
        ops.gdn_resident_blocks(200)
        assert ops.gdn_resident_blocks() == 200 and ops.gdn_resident_blocks() == 200, "the read must not drop an override (ADVICE r4)"
        ops.gdn_resident_blocks(-1)
        names = _gdn_kernel_names(lambda: run(True))
        assert len(names) == 1 and "gdn_chunk_single_kernel" in names[0] and "kernel<false, 1>" in names[0], names      # split pre-pass
        ops.gdn_resident_blocks(200)                                # 64 + 128 workgroups fit, 2 x 64 + 128 do not
        names = _gdn_kernel_names(lambda: same(run(True)) or pytest.fail("unsplit single launch differs"))
        assert len(names) == 1 and "kernel<false, 0>" in names[0], names
        ops.gdn_resident_blocks(100)                                # not even the scan workgroups fit: two launches
        names = _gdn_kernel_names(lambda: same(run(True)) or pytest.fail("two-launch fallback differs"))
        assert len(names) == 2 and any("prepare" in n for n in names) and any("scan_kernel" in n for n in names), names
        ops.gdn_resident_blocks(0)
        assert len(_gdn_kernel_names(lambda: run(True))) == 2
        # long call: needs twice the scan workgroups; with less it is prepare + scan per segment
        run_long = _gdn_fused_case(1000)
        ref_long = run_long(False)
        ops.gdn_resident_blocks(255)                                # 127 -> 112 persistent pre-pass workgroups (whole rounds of heads)
        assert same(run_long(True), ref_long)
        ops.gdn_resident_blocks(160)                                # 128 scan + 32 < 2 x 128
        names = _gdn_kernel_names(lambda: run_long(True))
        assert len(names) == 2 and not any("single" in n for n in names), names
        ops.gdn_resident_blocks(-1)
        names = _gdn_kernel_names(lambda: run_long(True))
        assert len(names) == 1 and "kernel<false, 2>" in names[0], names
    finally:
        ops.gdn_resident_blocks(restore)


@pytest.mark.parametrize("T,chunk", [(256, 1), (1000, 0), (1000, 10)])
def test_gdn_single_launch_reports_a_wait_that_runs_out(T, chunk):
    """A broken contract (here: a flag word that something else has written) must not produce silent garbage: the bounded wait
    runs out, the workgroups that waited store NOTHING (the head's final state keeps the caller's bytes), the failure is
    reported (ops.gdn_sync_check, and IVL_ERR_SYNC from the next call), queued launches stop at once, and after
    ops.gdn_sync_reset the area works again."""
    from infinitevl_amd import ops
    from infinitevl_amd._lib import IVL_ERR_SYNC, IvlError
    run = _gdn_fused_case(T)
    ref = run(False)
    area = ops.new_gdn_sync_area(DEV)
    try:
        with ops.gdn_sync_scope(area):
            assert torch.equal(run(True)[0], ref[0])
            torch.cuda.synchronize()
            ops.gdn_sync_check(DEV, deep=True)
            area.view(torch.int32)[64 * 2 + chunk] = 7            # head 2: the flag of `chunk` can never equal the producer count
            o, ht, _ = run(True, ht_fill=3.0)
            run(True)                                              # queued behind the failing launch: must give up at once
            torch.cuda.synchronize()
            assert bool((ht[0, 2] == 3.0).all()), "head 2 must not store a state computed from records it never saw"
            for h in (0, 1, 3, 15):
                assert torch.equal(ht[0, h], ref[1][0, h])         # the other heads are not affected
            with pytest.raises(IvlError) as ei:
                ops.gdn_sync_check(DEV)
            assert ei.value.code == IVL_ERR_SYNC and "head 2" in str(ei.value)
            with pytest.raises(IvlError):
                run(True)                                          # refused: nothing is launched on a failed area
            # ... while ANOTHER area (another stream's, another graph's) goes on: every area reports into a status slot of its own
            # (csrc/gdn_chunk.hip status_slot: handed out by registration since round 6)
            others = [ops.new_gdn_sync_area(DEV) for _ in range(24)]
            with ops.gdn_sync_scope(others[0]):
                ob = run(True)
                torch.cuda.synchronize()
                assert torch.equal(ob[0], ref[0]) and torch.equal(ob[1], ref[1])
            # re-arming OTHER areas (here 24 of them, one by one through the C entry point) must not clear the failed
            # area's report -- with slots assigned by address hash one of them could share its slot, and the next call on the failed
            # area was then launched and returned IVL_OK with incomplete outputs
            lib = ops._lib.load()
            for a in others:
                ops._lib.check(lib.ivl_gdn_sync_reset(a.data_ptr(), torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            with pytest.raises(IvlError) as ei2:
                ops.gdn_sync_check(DEV)
            assert ei2.value.code == IVL_ERR_SYNC
            with pytest.raises(IvlError):
                run(True)                                          # the failed area is still refused
            ops.gdn_sync_reset(DEV)
            ops.gdn_sync_check(DEV, deep=True)
            out = run(True)
            torch.cuda.synchronize()
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
            assert int(area.view(torch.int32).abs().sum()) == 0
    finally:
        torch.cuda.synchronize()
        ops.gdn_sync_reset(DEV)


def test_graphed_step_surfaces_a_failed_wait_and_recovers():
    """Harness level: a replayed graph whose single-launch GDN call cannot complete (its flag words were overwritten) does not
    hand back garbage silently -- GraphedStep.step() raises IvlError(IVL_ERR_SYNC) (at the failing step or the one after: the
    status word is read without a device synchronisation), and after ops.gdn_sync_reset() + a cache reset the same graph
    streams again and equals an undisturbed run."""
    from infinitevl_amd import ops
    from infinitevl_amd._lib import IVL_ERR_SYNC, IvlError
    from infinitevl_amd.harness import GraphedStep
    stack, hc, _, _ = _small_stack(window=96)
    T = 128
    xs = [bf(torch.randn(1, T, hc.hidden_size, device=DEV) * 0.5) for _ in range(3)]
    ref_cache = stack.allocate_inference_cache(1)
    with torch.no_grad():
        ref = [stack(inputs_embeds=x, past_key_values=ref_cache, logits_to_keep=1)[0].clone() for x in xs]
    cache = stack.allocate_inference_cache(1)
    gs = GraphedStep(stack, cache, 1, T, logits_to_keep=1)
    gs.capture()
    try:
        h0, _ = gs.step(xs[0])
        torch.cuda.synchronize()
        assert torch.equal(h0, ref[0])
        gs._sync.view(torch.int32)[0] = 9                      # head 0, chunk 0: can never equal the producer count
        with pytest.raises(IvlError) as ei:
            gs.step(xs[1])
            torch.cuda.synchronize()
            gs.step(xs[2])                                      # (the failure of the previous replay is visible by now at the latest)
        assert ei.value.code == IVL_ERR_SYNC
        torch.cuda.synchronize()
        ops.gdn_sync_reset(DEV)
        gs.reset()
        for x, r in zip(xs, ref):
            h, _ = gs.step(x)
            torch.cuda.synchronize()
            assert torch.equal(h, r)
    finally:
        torch.cuda.synchronize()
        ops.gdn_sync_reset(DEV)


def test_gdn_single_launch_under_a_co_running_stream():
    """Single-launch forms (step shape and long calls) while a second stream keeps the chip full of streaming kernels of uneven
    length: bit-equal to the two-launch form, no wait runs out, flags left cleared (tools/gdn_sync_stress.py)."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, os.path.join(root, "tools", "gdn_sync_stress.py"), "--T", "256,1000,4300", "--iters", "150", "--co-stream"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "GDN_SYNC_STRESS PASS" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_gdn_single_launch_two_processes_on_one_gpu():
    """Two processes on this one GPU, both issuing long single-launch calls (T = 1000 / 4300: 128 scan + 128 persistent pre-pass
    workgroups each, one CU per workgroup) and step-shape calls at the same time: neither can be resident in full.  A workgroup
    only waits for workgroups with lower block ids, so both make progress; results bit-equal to the two-launch form in both."""
    import subprocess
    import sys as _sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        bar = os.path.join(td, "go")
        procs = [subprocess.Popen([_sys.executable, os.path.join(root, "tools", "gdn_sync_stress.py"), "--T", "1000,4300,256", "--iters", "150",
                                   "--barrier-file", bar, "--nprocs", "2", "--seed", str(i)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(2)]
        outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "GDN_SYNC_STRESS PASS" in so, (so[-2000:], se[-2000:])


@pytest.mark.parametrize("M,N,K,glu,res", [(1, 12320, 2048, False, True), (1, 2560, 2048, False, True), (1, 11008, 2048, True, True),
                                           (4, 4096, 2048, False, False), (3, 1000, 4096, True, True), (2, 151936, 2048, False, True)])
def test_norm_linear_small_m_equals_the_separate_launches(M, N, K, glu, res):
    """ivl_norm_linear_small_m_fwd (the RMSNorm in the prologue of the decode step's weight stream) must equal
    ivl_add_rmsnorm_fwd followed by ivl_linear_small_m_fwd / ivl_linear_swiglu_small_m_fwd bit for bit: projection output and
    the new residual stream (std:1350-1429 at q_len == 1)."""
    from infinitevl_amd import ops
    g_ = torch.Generator(device=DEV).manual_seed(M * 7 + N)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    x, r = rn(1, M, K), (rn(1, M, K) if res else None)
    nw = bf(1.0 + 0.1 * torch.randn(K, device=DEV, generator=g_))
    W = rn(2 * N if glu else N, K) * 0.05
    bias = None if glu else rn(N)
    y_sep, h_sep = ops.add_rmsnorm(x, r, nw, 1e-6)
    out_sep = ops.linear_swiglu(y_sep, W) if glu else ops.linear(y_sep, W, bias)
    pn = ops.PreNorm(x, r, nw, 1e-6)
    out = ops.linear_swiglu(pn, W) if glu else ops.linear(pn, W, bias)
    assert pn.done and pn._y is None, "the fused launch must have been taken"
    assert torch.equal(out, out_sep)
    assert torch.equal(pn.h, h_sep)


def test_decode_step_with_norms_in_the_projection_prologue_is_bit_identical():
    """The decode step of the full-width stack with the 72 per-layer norms run inside the projection kernels (ops.PreNorm) must
    produce the logits and caches of the step with separate norm launches, bit for bit, eagerly and from the captured graph."""
    from infinitevl_amd import ops
    from infinitevl_amd.harness import GraphedDecode, InfiniteVLTextConfig, InfiniteVLTextStack
    cfg = InfiniteVLTextConfig(num_hidden_layers=8)
    stack = InfiniteVLTextStack(cfg).to(DEV).to(torch.bfloat16).init_weights_(seed=3).fuse_()
    g_ = torch.Generator(device=DEV).manual_seed(5)
    x = bf(torch.randn(1, 200, cfg.hidden_size, device=DEV, generator=g_) * 0.5)
    caches = []
    for flag in (False, True):
        ops._PRENORM = flag
        try:
            c = stack.allocate_inference_cache(1, zero_init=True)
            stack(inputs_embeds=x, past_key_values=c, logits_to_keep=1)
            lgs = []
            tok = torch.tensor([[17]], device=DEV)
            for it in range(6):
                _, lg = stack(input_ids=tok, past_key_values=c, logits_to_keep=1)
                lgs.append(lg.clone())
                tok = lg[:, -1].argmax(-1, keepdim=True)
            caches.append((c, lgs))
        finally:
            ops._PRENORM = True
    for a, b_ in zip(caches[0][1], caches[1][1]):
        assert torch.equal(a, b_)
    for la, lb in zip(caches[0][0].layers, caches[1][0].layers):
        for ta, tb in zip(la.carried_tensors(), lb.carried_tensors()):
            assert torch.equal(ta, tb)
    # graph replay of the fused form continues bit-identically to eager stepping of the separate form
    gd = GraphedDecode(stack, caches[1][0], 1)
    gd.token.fill_(23)
    tok = torch.tensor([[23]], device=DEV)
    ops._PRENORM = False
    try:
        for it in range(3):
            _, lg_ref = stack(input_ids=tok, past_key_values=caches[0][0], logits_to_keep=1)
            tok = lg_ref[:, -1].argmax(-1, keepdim=True)
            got = gd.step()
            assert torch.equal(gd.logits.reshape(-1), lg_ref.reshape(-1)), it
            assert torch.equal(got.reshape(-1), tok.reshape(-1))
    finally:
        ops._PRENORM = True


def test_rope_tables_kernel_matches_the_eager_chain():
    """ivl_rope_tables_fwd == the reference's eager rotary chain (std:896-930) on the same positions: identical bf16 tables
    (cos / sin in fp32 from the same fp32 product, rounded once), 3-D positions with distinct axes, large positions."""
    from infinitevl_amd import ops
    from infinitevl_amd.harness import InfiniteVLTextConfig
    from infinitevl_amd.modules import InfiniteVLRotaryEmbedding
    rot = InfiniteVLRotaryEmbedding(InfiniteVLTextConfig())
    B, T = 2, 300
    base = torch.tensor([0, 131072 - 17], device=DEV)[None, :, None]
    pos = torch.stack([torch.arange(T), torch.arange(T) // 3, torch.arange(T) % 7]).to(DEV)[:, None, :] + base
    inv = rot.inv_freq.to(DEV)
    freqs = pos[..., None].float() * inv
    emb = torch.cat((freqs, freqs), dim=-1)
    cos_ref, sin_ref = bf(emb.cos()), bf(emb.sin())
    cos, sin = ops.rope_tables(pos.contiguous(), inv)
    assert cos.shape == (3, B, T, 128) and cos.dtype == torch.bfloat16
    # the device cosf / sinf may differ from torch's by an fp32 ulp: at most one bf16 ulp on a handful of entries
    for got, ref in ((cos, cos_ref), (sin, sin_ref)):
        d = (got.float() - ref.float()).abs()
        assert float(d.max()) <= 2.0 ** -7 and float((d > 0).float().mean()) < 1e-2, (float(d.max()), float((d > 0).float().mean()))
    x = torch.zeros(1, dtype=torch.bfloat16, device=DEV)
    c2, s2 = rot(x, pos)
    assert torch.equal(c2, cos) and torch.equal(s2, sin)


@pytest.mark.parametrize("B,T,seen,W,rope", [(1, 256, 4500, 4096, True), (1, 256, 9000, 4096, False), (2, 1, 300, 96, True),
                                             (1, 4096, 100, 1024, False), (1, 70, 40, 96, True), (1, 5000, 0, 4096, False)])
def test_swa_append_folded_into_the_call_equals_separate_append(B, T, seen, W, rope):
    """ivl_swa_args.append_new (the ring append riding in the split-KV combine launch, or launched by ivl_swa_fwd itself when
    there is no combine) must leave the ring exactly as ivl_swa_cache_append does (reference tail copy-back, std:146-172):
    split and unsplit prefill, decode, T > capacity, with and without the fused rotation of the keys."""
    from infinitevl_amd import ops
    Hq, Hkv, d, C = 16, 2, 128, W - 1
    g_ = torch.Generator(device=DEV).manual_seed(T + seen)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    q, k, v = rn(B, T, Hq, d), rn(B, T, Hkv, d), rn(B, T, Hkv, d)
    kc, vc = rn(B, Hkv, C, d), rn(B, Hkv, C, d)
    pos_dev = torch.full((1,), seen, dtype=torch.int64, device=DEV)
    rp = None
    if rope:
        cos, sin = rn(3, B, T, d), rn(3, B, T, d)
        rp = (cos, sin, [16, 24, 24])
    kc1, vc1, kc2, vc2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    o1 = ops.swa_forward(q, k, v, window=W, scaling=d ** -0.5, k_cache=kc1, v_cache=vc1, pos_dev=pos_dev, rope=rp)
    ops.swa_cache_append(k, v, kc1, vc1, pos_dev=pos_dev, rope=rp)
    o2 = ops.swa_forward(q, k, v, window=W, scaling=d ** -0.5, k_cache=kc2, v_cache=vc2, pos_dev=pos_dev, rope=rp, append=True)
    assert torch.equal(o1, o2)
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    assert not torch.equal(kc, kc2)


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[1] / configs[3] bulk prefill: the SWA product path of a LONG call (fused M-RoPE at T >= 512:
# rope pre-pass -> swa_prefill_kernel on the rotated workspace copies -> ring append launched by ivl_swa_fwd itself or riding
# in the split-KV combine), value-checked at the real head shape
# ---------------------------------------------------------------------------------------------
def _mrope_tables(B, T, seen, d=128):
    from infinitevl_amd.harness import InfiniteVLTextConfig
    from infinitevl_amd.modules import InfiniteVLRotaryEmbedding
    pos = torch.stack([torch.arange(seen, seen + T), torch.arange(T) // 3 + seen, torch.arange(T) % 7 + seen])[:, None, :]
    pos = pos.expand(3, B, T).to(DEV)
    x = torch.zeros(1, dtype=torch.bfloat16, device=DEV)
    return InfiniteVLRotaryEmbedding(InfiniteVLTextConfig())(x, pos)


BULK_CASES = [(B, T, ring, W) for W in (4096, 1024) for T in (512, 1000, 4096) for ring in ("empty", "half", "wrapped") for B in (1,)]
BULK_CASES += [(2, 1000, "wrapped", 4096), (2, 2048, "half", 4096), (1, 2500, "wrapped", 4096)]      # single split below 4096 tokens


@pytest.mark.parametrize("B,T,ring,W", BULK_CASES)
def test_swa_bulk_prefill_with_fused_rope_and_append_is_bit_identical(B, T, ring, W):
    """ops.swa_forward(..., rope=(cos, sin, [16,24,24]), append=True) at T in {512, 1000, 4096} (split and single-split
    launches of swa_prefill_kernel behind the rope pre-pass) over an empty, a half-full and a wrapped full ring must equal
    apply_mrope_inplace + plain swa_forward + swa_cache_append bit for bit, in the output AND in the ring."""
    from infinitevl_amd import ops
    Hq, Hkv, d, C = 16, 2, 128, W - 1
    seen = {"empty": 0, "half": C // 2, "wrapped": 3 * C + 17}[ring]
    g_ = torch.Generator(device=DEV).manual_seed(T + seen + B)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    q, k, v = rn(B, T, Hq, d), rn(B, T, Hkv, d), rn(B, T, Hkv, d)
    kc, vc = rn(B, Hkv, C, d), rn(B, Hkv, C, d)
    cos, sin = _mrope_tables(B, T, seen)
    sec = [16, 24, 24]
    pos_dev = torch.full((1,), seen, dtype=torch.int64, device=DEV)
    # rotate, attend, append: three plain calls
    q1, k1 = q.clone(), k.clone()
    ops.apply_mrope_inplace(q1, k1, cos, sin, sec)
    kc1, vc1 = kc.clone(), vc.clone()
    o1 = ops.swa_forward(q1, k1, v, window=W, scaling=d ** -0.5, k_cache=kc1, v_cache=vc1, pos_dev=pos_dev)
    ops.swa_cache_append(k1, v, kc1, vc1, pos_dev=pos_dev)
    # the product call
    kc2, vc2 = kc.clone(), vc.clone()
    o2 = ops.swa_forward(q, k, v, window=W, scaling=d ** -0.5, k_cache=kc2, v_cache=vc2, pos_dev=pos_dev, rope=(cos, sin, sec),
                         append=True)
    torch.cuda.synchronize()
    assert torch.isfinite(o2.float()).all()
    assert torch.equal(o1, o2)
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    assert not torch.equal(kc, kc2)
    # the ring holds the LAST min(T, C) rotated keys at slots (seen + t) % C (reference tail copy-back, std:146-172)
    t_first = max(0, T - C)
    slots = (seen + torch.arange(t_first, T, device=DEV)) % C
    assert torch.equal(kc2[:, :, slots], k1[:, t_first:].transpose(1, 2))
    assert torch.equal(vc2[:, :, slots], v[:, t_first:].transpose(1, 2))


@pytest.mark.parametrize("T,seen,W,Hq,Hkv", [(512, 0, 4096, 2, 1), (512, 700, 1024, 2, 1), (512, 3000, 1024, 4, 2), (1000, 5000, 4096, 2, 1)])
def test_swa_bulk_prefill_with_fused_rope_vs_oracle(T, seen, W, Hq, Hkv):
    """The same product call against the CPU oracle: M-RoPE in fp32 on the bf16 inputs / tables (q, k rounded to bf16 behind
    it: the reference's rotated q / k are bf16 tensors, std:1057-1064), exact softmax attention on the S2 band: <= 5e-3."""
    from infinitevl_amd import ops
    B, d, C = 1, 128, W - 1
    g_ = torch.Generator().manual_seed(T + seen)
    sn = lambda *sh: torch.randn(*sh, generator=g_).to(torch.bfloat16)      # noqa: E731
    q, k, v = sn(B, T, Hq, d), sn(B, T, Hkv, d), sn(B, T, Hkv, d)
    n_prev = oswa.n_prev_keys(W, seen)
    k_hist, v_hist = sn(B, seen, Hkv, d), sn(B, seen, Hkv, d)               # the ring's keys: rotated earlier
    cos, sin = _mrope_tables(B, T, seen)
    sec = [16, 24, 24]
    qr, kr = oswa.apply_mrope(q.float().transpose(1, 2), k.float().transpose(1, 2), cos.float().cpu(), sin.float().cpu(), sec)
    qr, kr = qr.to(torch.bfloat16).float(), kr.to(torch.bfloat16).float()
    k_all = torch.cat([k_hist[:, seen - n_prev:].float().transpose(1, 2), kr], dim=2)
    v_all = torch.cat([v_hist[:, seen - n_prev:].float(), v.float()], dim=1).transpose(1, 2)
    ref = oswa.swa_attention(qr, k_all, v_all, n_prev, W, d ** -0.5)
    kc = torch.zeros(B, Hkv, C, d, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    pos_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
    for a in range(0, seen, 333):
        n = min(333, seen - a)
        ops.swa_cache_append(k_hist[:, a:a + n].to(DEV), v_hist[:, a:a + n].to(DEV), kc, vc, pos_dev=pos_dev)
        ops.counter_add(pos_dev, n)
    out = ops.swa_forward(q.to(DEV), k.to(DEV), v.to(DEV), window=W, scaling=d ** -0.5, k_cache=kc, v_cache=vc, pos_dev=pos_dev,
                          rope=(cos, sin, sec), append=True)
    torch.cuda.synchronize()
    err = rms_rel(ref, out.float().cpu())
    assert err < 5e-3, err
    # and the appended keys are the oracle's rotated keys (the rotation's three roundings vs one: <= 1 bf16 ulp apart)
    t_first = max(0, T - C)
    slots = (seen + torch.arange(t_first, T)) % C
    got = kc[:, :, slots.to(DEV)].float().cpu()
    assert rms_rel(kr[:, :, t_first:], got) < 4e-3


def _swa_random_case(seed):
    import random
    r = random.Random(seed)
    W = r.choice([64, 96, 500, 1024, 4096])
    C = W - 1
    Hq, Hkv = r.choice([(2, 1), (4, 2), (16, 2)])
    B = r.choice([1, 1, 2])
    T = r.choice([r.randint(1, 70), r.randint(65, 700), r.randint(500, 1400)])
    seen = r.choice([0, r.randint(1, C), r.randint(C, 4 * C), C, 2 * C - 1, r.randint(0, 3) * C + r.randint(0, 130)])
    if Hq == 16:
        T = min(T, 600)                                                    # keeps the CPU oracle at seconds
    return B, T, Hq, Hkv, W, seen, r.random() < 0.5, r.random() < 0.6


@pytest.mark.parametrize("seed", range(int(os.environ.get("IVL_SWA_FUZZ", "32"))))
def test_swa_random_shapes_vs_oracle(seed):
    """Randomised differential test of `ops.swa_forward` (every launch form: packed rows, 64- and 128-row workgroups, split and
    single-split, ring / new / seam / wrap / tail tiles, with and without the fused M-RoPE and the folded append) against the
    CPU oracle: ragged T, windows from 64 to 4096, rings empty / partly filled / wrapped several times.  IVL_SWA_FUZZ=N runs N
    seeds (default 32)."""
    from infinitevl_amd import ops
    B, T, Hq, Hkv, W, seen, rope, append = _swa_random_case(seed)
    d, C = 128, W - 1
    g_ = torch.Generator().manual_seed(1000 + seed)
    sn = lambda *sh: torch.randn(*sh, generator=g_).to(torch.bfloat16)      # noqa: E731
    q, k, v = sn(B, T, Hq, d), sn(B, T, Hkv, d), sn(B, T, Hkv, d)
    n_prev = oswa.n_prev_keys(W, seen)
    hist = min(seen, 2 * C + 50)                                            # keys older than the ring's capacity are gone anyway
    k_hist, v_hist = sn(B, hist, Hkv, d), sn(B, hist, Hkv, d)
    qr, kr = q.float().transpose(1, 2), k.float().transpose(1, 2)
    sec = [16, 24, 24]
    if rope:
        cos, sin = _mrope_tables(B, T, seen)
        qr, kr = oswa.apply_mrope(qr, kr, cos.float().cpu(), sin.float().cpu(), sec)
        qr, kr = qr.to(torch.bfloat16).float(), kr.to(torch.bfloat16).float()
    k_all = torch.cat([k_hist[:, hist - n_prev:].float().transpose(1, 2), kr], dim=2)
    v_all = torch.cat([v_hist[:, hist - n_prev:].float(), v.float()], dim=1).transpose(1, 2)
    ref = oswa.swa_attention(qr, k_all, v_all, n_prev, W, d ** -0.5)
    kc = torch.zeros(B, Hkv, C, d, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    pos_dev = torch.full((1,), seen - hist, dtype=torch.int64, device=DEV)
    for a in range(0, hist, 333):
        n = min(333, hist - a)
        ops.swa_cache_append(k_hist[:, a:a + n].to(DEV), v_hist[:, a:a + n].to(DEV), kc, vc, pos_dev=pos_dev)
        ops.counter_add(pos_dev, n)
    kc0, vc0 = kc.clone(), vc.clone()
    out = ops.swa_forward(q.to(DEV), k.to(DEV), v.to(DEV), window=W, scaling=d ** -0.5, k_cache=kc, v_cache=vc, pos_dev=pos_dev,
                          rope=(cos, sin, sec) if rope else None, append=append)
    torch.cuda.synchronize()
    case = dict(B=B, T=T, Hq=Hq, Hkv=Hkv, W=W, seen=seen, rope=rope, append=append)
    assert torch.isfinite(out.float()).all(), case
    err = rms_rel(ref, out.float().cpu())
    assert err < 5e-3, (err, case)
    # per-row check: a single wrong row in a long call does not move the RMS over the whole tensor
    row = ((ref - out.float().cpu()) ** 2).sum(-1).sqrt() / (ref ** 2).sum(-1).sqrt().clamp_min(1e-3)
    assert float(row.max()) < 4e-2, (float(row.max()), case)
    if append:
        t_first = max(0, T - C)
        slots = (seen + torch.arange(t_first, T)) % C
        got = kc[:, :, slots.to(DEV)].float().cpu()
        assert rms_rel(kr[:, :, t_first:], got) < 4e-3, case
        assert torch.equal(vc[:, :, slots.to(DEV)].cpu(), v[:, t_first:].transpose(1, 2)), case
        keep = torch.ones(C, dtype=torch.bool)
        keep[slots] = False                                                 # every other slot is untouched
        assert torch.equal(kc[:, :, keep.to(DEV)], kc0[:, :, keep.to(DEV)]) and torch.equal(vc[:, :, keep.to(DEV)], vc0[:, :, keep.to(DEV)]), case
    else:
        assert torch.equal(kc, kc0) and torch.equal(vc, vc0), case


def test_swa_one_4096_token_call_equals_sixteen_256_token_calls():
    """configs[1]'s prefill form against configs[2]'s streaming form at the operator: the ring after ONE 4096-token product
    call is bit-equal to the ring after sixteen 256-token product calls (the rotation is element-wise and the append places
    token t at (pos + t) % C either way); the outputs are two bf16 results of the same rows -- one pass per row in the long call,
    eight bf16 partial rows merged in the 256-token calls -- each within the operator's tolerance of the exact result, <= 5e-3
    from each other (observed 3.5e-3)."""
    from infinitevl_amd import ops
    B, T, Hq, Hkv, d, W = 1, 4096, 16, 2, 128, 4096
    C, seen = W - 1, 2 * (W - 1) + 99
    g_ = torch.Generator(device=DEV).manual_seed(5)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    q, k, v = rn(B, T, Hq, d), rn(B, T, Hkv, d), rn(B, T, Hkv, d)
    kc, vc = rn(B, Hkv, C, d), rn(B, Hkv, C, d)
    cos, sin = _mrope_tables(B, T, seen)
    sec = [16, 24, 24]
    kc1, vc1 = kc.clone(), vc.clone()
    p1 = torch.full((1,), seen, dtype=torch.int64, device=DEV)
    o1 = ops.swa_forward(q, k, v, window=W, scaling=d ** -0.5, k_cache=kc1, v_cache=vc1, pos_dev=p1, rope=(cos, sin, sec), append=True)
    kc2, vc2 = kc.clone(), vc.clone()
    p2 = torch.full((1,), seen, dtype=torch.int64, device=DEV)
    outs = []
    for a in range(0, T, 256):
        sl = slice(a, a + 256)
        outs.append(ops.swa_forward(q[:, sl], k[:, sl], v[:, sl], window=W, scaling=d ** -0.5, k_cache=kc2, v_cache=vc2, pos_dev=p2,
                                    rope=(cos[:, :, sl].contiguous(), sin[:, :, sl].contiguous(), sec), append=True))
        ops.counter_add(p2, 256)
    o2 = torch.cat(outs, dim=1)
    torch.cuda.synchronize()
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    assert rms_rel(o1.float(), o2.float()) < 5e-3, rms_rel(o1.float(), o2.float())


def _full_size_stack(window=4096, seed=0):
    from infinitevl_amd.harness import InfiniteVLTextConfig, InfiniteVLTextStack
    cfg = InfiniteVLTextConfig(sliding_window=window)
    with torch.device(DEV):
        torch.set_default_dtype(torch.bfloat16)
        try:
            model = InfiniteVLTextStack(cfg)
        finally:
            torch.set_default_dtype(torch.float32)
    model = model.to(torch.bfloat16).eval()
    model.init_weights_(seed=seed).fuse_()
    return cfg, model


def test_configs1_full_size_4096_token_call_vs_streaming_then_128_graphed_decode_steps():
    """BASELINE.json configs[1] (InfiniteVL-3B shape, 36 layers, bf16: 4096-token prefill in ONE call + 128 decode steps) on the
    product path, value-checked against configs[2]'s streaming form (sixteen 256-token calls: an independent launch
    configuration of every kernel -- single-split attention behind the rope pre-pass vs 8-way split KV, long-call GDN launch vs
    step-shape launch, M = 4096 vs M = 256 library GEMMs):
    (i)   layer by layer on IDENTICAL inputs: the mixer input of every decoder layer of the one-call run is replayed through the
          same mixer in sixteen 256-token calls on a fresh cache -- SWA rings and conv states bit-equal, mixer outputs and the
          (bf16-carried) GDN states within the tolerance of two bf16 results of one operator;
    (ii)  end to end: the one-call run differs from the streamed run by no more than two streamed runs (256- vs 128-token
          calls) differ from each other (the bf16 noise of 36 random-weight layers, measured in the same test), x1.5;
    (iii) 128 GraphedDecode steps from the prefilled cache equal the eager greedy loop from a clone of it bit for bit."""
    import gc
    from infinitevl_amd.harness import GraphedDecode, clone_inference_cache, greedy_decode
    cfg, model = _full_size_stack()
    T = 4096
    g_ = torch.Generator(device=DEV).manual_seed(11)
    x = (torch.randn(1, T, cfg.hidden_size, device=DEV, generator=g_) * 0.02).to(torch.bfloat16)
    ins, outs, hooks = {}, {}, []
    for layer in model.layers:
        i = layer.self_attn.layer_idx
        hooks.append(layer.self_attn.register_forward_pre_hook(
            lambda m, a, kw, i=i: ins.__setitem__(i, kw["hidden_states"].clone()), with_kwargs=True))
        hooks.append(layer.self_attn.register_forward_hook(lambda m, a, o, i=i: outs.__setitem__(i, o[0].clone())))
    with torch.no_grad():
        ca = model.allocate_inference_cache(1)
        ha, la = model(inputs_embeds=x, past_key_values=ca)
        for h in hooks:
            h.remove()
        torch.cuda.synchronize()
        assert ca.get_seq_length() == T and torch.isfinite(ha.float()).all() and len(ins) == len(outs) == 36
        # (i) every layer's bulk call against its own streaming form on the same input
        cr = model.allocate_inference_cache(1)
        pos_all = torch.arange(T, device=DEV)[None, None, :].expand(3, 1, T)
        worst = {"swa_out": 0.0, "gdn_out": 0.0, "ring": 0.0, "state": 0.0, "conv": 0.0, "rings_bit_equal": 0}
        for layer in model.layers:
            i = layer.self_attn.layer_idx
            l1, l2 = ca.layers[i], cr.layers[i]
            sliding = getattr(l2, "is_sliding", False)
            if sliding:
                l2._advances_counter = True                    # replayed alone: this layer advances the (shared) position counter
                l2._pos_dev.zero_()
            ys = []
            for a in range(0, T, 256):
                pid = pos_all[:, :, a:a + 256].contiguous()
                pe = model.rotary_emb(x, pid)
                y, _ = layer.self_attn(hidden_states=ins[i][:, a:a + 256], position_ids=pid, past_key_values=cr, use_cache=True,
                                       position_embeddings=pe)
                ys.append(y)
            e_out = rms_rel(outs[i].float(), torch.cat(ys, dim=1).float())
            if sliding:
                assert int(l2._pos_dev.item()) == T and l1.size == l2.size == min(T, l1.capacity)
                ek, ev = rms_rel(l1._buf_keys.float(), l2._buf_keys.float()), rms_rel(l1._buf_values.float(), l2._buf_values.float())
                worst["ring"] = max(worst["ring"], ek, ev)
                worst["rings_bit_equal"] += int(torch.equal(l1._buf_keys, l2._buf_keys) and torch.equal(l1._buf_values, l2._buf_values))
                worst["swa_out"] = max(worst["swa_out"], e_out)
            else:
                worst["gdn_out"] = max(worst["gdn_out"], e_out)
                worst["state"] = max(worst["state"], rms_rel(l1.recurrent_state.float(), l2.recurrent_state.float()))
                for nm in ("conv_state_q", "conv_state_k", "conv_state_v"):
                    worst["conv"] = max(worst["conv"], rms_rel(getattr(l1, nm).float(), getattr(l2, nm).float()))
        print("configs[1] per-layer bulk vs streaming on identical inputs:", worst)
        # two bf16 results of one operator (5e-3 each from exact by the operator tolerance); rotated keys / values / conv inputs
        # differ by the rounding of the M = 4096 vs M = 256 projection GEMM only; the GDN state is carried in bf16 (Q5): 16
        # roundings of the carried state on the streaming side against 1
        # (observed: outputs 3.0e-3 / 3.1e-3, state 3.0e-3; the rings of all 9 sliding layers and every conv state BIT-EQUAL: the
        # library GEMM is row-stable between M = 4096 and M = 256 on this stack, the rotation and the append are element-wise)
        assert worst["swa_out"] < 6e-3 and worst["gdn_out"] < 6e-3 and worst["state"] < 6e-3, worst
        assert worst["rings_bit_equal"] == 9 and worst["ring"] == 0.0 and worst["conv"] == 0.0, worst
        del ins, outs, cr
        # (ii) end to end against the noise floor of the streaming form itself
        def streamed(step):
            c_ = model.allocate_inference_cache(1)
            hs = [model(inputs_embeds=x[:, a:a + step], past_key_values=c_)[0] for a in range(0, T, step)]
            return torch.cat(hs, dim=1), c_
        hb, cb = streamed(256)
        hc, cc = streamed(128)
        torch.cuda.synchronize()
        floor = rms_rel(hb.float(), hc.float())
        e_h = rms_rel(hb.float(), ha.float())
        print(f"configs[1] end to end: one call vs 16 x 256: {e_h:.4f}; 16 x 256 vs 32 x 128 (noise floor): {floor:.4f}")
        # (the bound was 1.5 x floor + 1e-2.)  Both numbers are distances between two bf16 trajectories of the same 36
        # random-weight layers (hidden rms-relative ~9.5e-2: the amplification of 36 layers, not of one operator -- (i) holds every
        # layer to 6e-3 on identical inputs); observed 0.0987 against a floor of 0.0949, ratio 1.04.  Two such distances agree to
        # ~10 %, so the one-call form is held to 1.2 x the floor + 2e-3.  What this half can see: an error of the bulk path that adds
        # in quadrature to the floor must reach ~0.065 end to end, i.e. ~1.1e-2 per layer over 36 layers (two operator tolerances),
        # to fail it -- (i) is the tight half (6e-3 per layer on identical inputs), (ii) guards the composition.
        assert torch.isfinite(hb.float()).all() and e_h < 1.2 * floor + 2e-3 and e_h < 0.2, (e_h, floor)
        for l1, l2 in zip(ca.layers, cb.layers):
            if getattr(l1, "is_sliding", False):
                assert int(l1._pos_dev.item()) == int(l2._pos_dev.item()) == T
        del hb, hc, cb, cc
        # (iii) decode: graph replays == eager loop, bit for bit
        first = la[:, -1].argmax(-1)
        ce = clone_inference_cache(ca)
        toks_e = greedy_decode(model, ce, first, steps=128)
        _, lg_e = model(input_ids=toks_e[:, -1:], past_key_values=ce)
        gd = GraphedDecode(model, ca, 1)
        gd.token.copy_(first.view(1, 1))
        toks_g = []
        for _ in range(128):
            toks_g.append(gd.step().clone())
        toks_g = torch.cat(toks_g, dim=1)
        gd.step()
        torch.cuda.synchronize()
        assert torch.equal(toks_e, toks_g)
        assert torch.equal(lg_e, gd.logits)
        assert ca.get_seq_length() == ce.get_seq_length() == T + 129
        for l1, l2 in zip(ca.layers, ce.layers):
            if getattr(l1, "is_sliding", False):
                assert torch.equal(l1._buf_keys, l2._buf_keys) and torch.equal(l1._buf_values, l2._buf_values)
            else:
                assert torch.equal(l1.recurrent_state, l2.recurrent_state)
    del model, ca, ce, gd
    gc.collect()
    torch.cuda.empty_cache()


_RCCL_WORLD1 = r"""
import json, os, sys, torch
sys.path.insert(0, os.environ["IVL_ROOT"])
import torch.distributed as dist
from infinitevl_amd import dist as idist
rank, world, local = idist.init_distributed("nccl", force=True)
dev = torch.device("cuda", local)
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
g = torch.Generator(device=dev).manual_seed(1)
lg = torch.randn(3, 151936, device=dev, generator=g).to(torch.bfloat16)
got = idist.gather_last_logits(lg, [3])
out["gather_equal"] = bool(torch.equal(got, lg)) and got.data_ptr() != lg.data_ptr()
out["max"] = idist.max_over_ranks(1.25, dev)
out["ranks"] = idist.describe_ranks(dev)
idist.barrier()
# the point-to-point primitive of the sequence-parallel hand-off on DEVICE tensors (dist.py: batch_isend_irecv), to self
try:
    a = torch.arange(1 << 20, device=dev, dtype=torch.float32)
    b = torch.zeros_like(a)
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)])
    for r_ in reqs:
        r_.wait()
    torch.cuda.synchronize()
    out["p2p_self"] = bool(torch.equal(a, b))
except Exception as e:                                  # recorded, not fatal: self send / recv is an RCCL capability, not ours
    out["p2p_self"] = "error: " + repr(e)[:200]
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_rccl_first_contact_world_size_one():
    """backend="nccl" (= RCCL on ROCm) at world size 1 on cuda:0, in a process of its own: librccl loads, the `device_id=` init
    path and HSA_ENABLE_IPC_MODE_LEGACY=0 work on the box, and the collectives of the multi-GPU path (SURVEY.md 8e:
    gather_last_logits' all_gather_into_tensor, max_over_ranks' all_reduce, describe_ranks' all_gather_object, barrier) run on
    device tensors and return the input -- the cheapest things an 8-GPU run can fail on before it measures anything."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                "HSA_ENABLE_IPC_MODE_LEGACY": "0", "IVL_ROOT": os.path.dirname(os.path.dirname(os.path.abspath(__file__)))})
    r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["backend"] == "nccl" and out["world"] == 1 and out["ipc_legacy"] == "0", out
    assert out["gather_equal"] is True and out["max"] == 1.25, out
    assert out["ranks"]["world"] == 1 and out["ranks"]["distinct_devices"] == 1 and out["ranks"]["devices"][0]["device_index"] == 0, out
    print("rccl world-1:", out)
    assert out["p2p_self"] is True or str(out["p2p_self"]).startswith("error"), out


def test_gdn_concurrent_streams_and_graphs_have_their_own_records():
    """the chunk-record workspace is keyed like the sync area -- per stream for eager calls, owned by the
    scope / graph for captures.  Two streams issuing long fused GDN calls on DIFFERENT inputs at the same time, and two captured
    graphs replayed at the same time on two streams, must each return exactly what they return alone (with one records buffer per
    device the pre-pass of one launch overwrites records the other's scan has already seen flagged: silently wrong outputs)."""
    from infinitevl_amd import ops
    runs = [_gdn_fused_case(T, seed=sd) for T, sd in ((4096, 1), (4096, 2))]
    refs = [r(True) for r in runs]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for it in range(12):
        outs = [None, None]
        for i, st in enumerate((s1, s2)):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for _ in range(3):
                    outs[i] = runs[i](True)
        torch.cuda.synchronize()
        for (o, h, so), (ro, rh, rso) in zip(outs, refs):
            assert torch.equal(o, ro) and torch.equal(h, rh), it
            assert all(torch.equal(a, b) for a, b in zip(so, rso)), it
    a1 = ops._gdn_sync_area(DEV)
    with torch.cuda.stream(s1):
        w1 = ops._gdn_workspace(1, ops._gdn_sync_area(DEV))
    with torch.cuda.stream(s2):
        w2 = ops._gdn_workspace(1, ops._gdn_sync_area(DEV))
    assert w1.data_ptr() != w2.data_ptr() and a1.data_ptr() not in (w1.data_ptr(), w2.data_ptr())
    # two graphs, each capturing a fused call inside a scope of its own, replayed concurrently
    graphs, gouts = [], []
    for i, st in enumerate((s1, s2)):
        area = ops.new_gdn_sync_area(DEV)
        with ops.gdn_sync_scope(area):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                runs[i](True)                                    # warm-up: creates the area's workspace outside the capture
            torch.cuda.current_stream().wait_stream(st)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                gouts.append(runs[i](True))
        graphs.append((g, area))
    assert ops._gdn_workspace(1, graphs[0][1]).data_ptr() != ops._gdn_workspace(1, graphs[1][1]).data_ptr()
    for it in range(12):
        for (g, _), st in zip(graphs, (s1, s2)):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                g.replay()
                g.replay()
        torch.cuda.synchronize()
        for (o, h, so), (ro, rh, rso) in zip(gouts, refs):
            assert torch.equal(o, ro) and torch.equal(h, rh), it
    ops.gdn_sync_check(DEV, deep=True)


def test_swa_prefill_is_bit_stable_under_a_co_running_stream():
    """The prefill kernel's tile ring is filled by LDS-DMA that runs up to three tiles ahead of its readers; every read is ordered
    behind the landing of its tile by a counted vmcnt + a workgroup barrier, never by timing.  (A version whose rotated key half
    read a tile one barrier before its landing was guaranteed passed every quiet test and failed 4 of 12 two-process runs.)  So:
    long and step-shape calls while a second stream keeps the memory system busy with 1 GiB copies -- every output and ring
    equals the quiet run's, bit for bit, 40 times.  (Attention calls on ONE stream: concurrent calls would share the package's
    per-device "swa" workspace, which ops.get_workspace documents as one stream at a time.)"""
    from infinitevl_amd import ops
    Hq, Hkv, d, W = 16, 2, 128, 4096
    C = W - 1
    g_ = torch.Generator(device=DEV).manual_seed(77)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    cases = []
    for T, seen in ((4096, 3 * C + 5), (256, 2 * C + 100), (1000, 17)):
        q, k, v = rn(1, T, Hq, d), rn(1, T, Hkv, d), rn(1, T, Hkv, d)
        kc, vc = rn(1, Hkv, C, d), rn(1, Hkv, C, d)
        cos, sin = _mrope_tables(1, T, seen)
        pos_dev = torch.full((1,), seen, dtype=torch.int64, device=DEV)

        def run(q=q, k=k, v=v, kc=kc, vc=vc, cos=cos, sin=sin, pos_dev=pos_dev):
            kc2, vc2 = kc.clone(), vc.clone()
            o = ops.swa_forward(q, k, v, window=W, scaling=d ** -0.5, k_cache=kc2, v_cache=vc2, pos_dev=pos_dev,
                                rope=(cos, sin, [16, 24, 24]), append=True)
            return o, kc2, vc2
        cases.append((run, run()))
    torch.cuda.synchronize()
    a_, b_ = torch.empty(1 << 28, dtype=torch.float32, device=DEV), torch.empty(1 << 28, dtype=torch.float32, device=DEV)
    noise = torch.cuda.Stream()
    for it in range(40):
        with torch.cuda.stream(noise):
            b_.copy_(a_)
            a_.copy_(b_)
        outs = [run() for run, _ in cases]
        torch.cuda.synchronize()
        for (o, kc2, vc2), (_, (ro, rk, rv)) in zip(outs, cases):
            assert torch.equal(o, ro) and torch.equal(kc2, rk) and torch.equal(vc2, rv), it


@pytest.mark.parametrize("B,H,state_dtype", [(1, 16, torch.bfloat16), (2, 16, torch.float32), (4, 16, torch.bfloat16), (3, 5, torch.bfloat16)])
def test_gdn_decode_split_plus_out_linear_equals_one_launch_step_plus_linear(B, H, state_dtype):
    """Round 5: the decode step on 64 workgroups (sequence x head x quarter of the value columns,
    ivl_gdn_decode_split_fwd) with the gated RMSNorm and the q / k conv-state shift in the prologue of the o_proj weight stream
    (ivl_gdn_out_linear_small_m_fwd) equals ivl_gdn_decode_step_fwd + ivl_linear_small_m_fwd BIT FOR BIT over several steps: the
    projected output, the recurrent state and all three conv states (same 16-row fma chains and group order in the column sums,
    the norm on the same lane -> element map)."""
    from infinitevl_amd import ops
    K, V = 128, 256
    Dq, Dv = H * K, H * V
    cols = (0, Dq, 2 * Dq, 2 * Dq + Dv, 2 * Dq + 2 * Dv, 2 * Dq + 2 * Dv + H)      # q, k, v, g, a, b
    ld = cols[5] + H
    ld += (-ld) % 8
    g_ = torch.Generator(device=DEV).manual_seed(B * 100 + H)
    rn = lambda *sh: bf(torch.randn(*sh, device=DEV, generator=g_))      # noqa: E731
    state0 = rn(B, H, K, V).to(state_dtype)
    conv0 = [rn(B, D, 4) for D in (Dq, Dq, Dv)]
    cw = [rn(D, 1, 4) * 0.5 for D in (Dq, Dq, Dv)]
    A32, dt32 = torch.randn(H, device=DEV, generator=g_), torch.randn(H, device=DEV, generator=g_)
    wn = rn(V)
    N = 2048
    wo, bo = rn(N, Dv) * 0.05, (rn(N) if H == 5 else None)
    st_a, st_b = state0.clone(), state0.clone()
    ca, cb = [c.clone() for c in conv0], [c.clone() for c in conv0]
    for step in range(4):
        proj = rn(B, 1, ld)
        y1 = ops.gdn_decode_step(proj, cols, cw, ca, A32, dt32, wn, 1e-5, st_a, H, K, V, K ** -0.5)
        o1 = ops.linear(y1, wo, bo)
        o_raw = ops.gdn_decode_split(proj, (cols[0], cols[1], cols[2], cols[4], cols[5]), cw, cb, A32, dt32, st_b, H, K, V, K ** -0.5)
        o2 = ops.gdn_out_linear(o_raw, proj, cols[3], cols[0], cols[1], wn, 1e-5, cb[0], cb[1], wo, bo, H)
        torch.cuda.synchronize()
        assert torch.equal(o1, o2), step
        assert torch.equal(st_a, st_b), step
        for x_a, x_b in zip(ca, cb):
            assert torch.equal(x_a, x_b), step
    assert torch.isfinite(o2.float()).all() and not torch.equal(st_a, state0)


def test_decode_tokens_with_the_split_step_equal_the_one_launch_step():
    """Module level: greedy decode through the 4-layer stack with ops._SPLIT_DECODE on / off: tokens, logits and caches bit-equal."""
    from infinitevl_amd import ops
    from infinitevl_amd.harness import greedy_decode
    stack, hc, _, _ = _small_stack(window=96)
    x = bf(torch.randn(1, 70, hc.hidden_size, device=DEV) * 0.5)
    res = []
    for flag in (True, False):
        ops._SPLIT_DECODE = flag
        try:
            with torch.no_grad():
                cache = stack.allocate_inference_cache(1)
                _, lg = stack(inputs_embeds=x, past_key_values=cache)
                toks = greedy_decode(stack, cache, lg[:, -1].argmax(-1), steps=12)
                _, lg2 = stack(input_ids=toks[:, -1:], past_key_values=cache)
            res.append((toks, lg2, cache))
        finally:
            ops._SPLIT_DECODE = False
    (t1, l1, c1), (t2, l2, c2) = res
    assert torch.equal(t1, t2) and torch.equal(l1, l2)
    for a, b_ in zip(c1.layers, c2.layers):
        if not getattr(a, "is_sliding", False):
            assert torch.equal(a.recurrent_state, b_.recurrent_state)
            for nm in ("conv_state_q", "conv_state_k", "conv_state_v"):
                assert torch.equal(getattr(a, nm), getattr(b_, nm)), nm


def test_gdn_workspace_mirror_is_capped_and_prepare_release_work():
    """an eager scope-less call no longer grows the records workspace of the device's area for scope-less CAPTURES beyond
    16 MB (a 4096-token call is ~64 MB: plain eager use held it twice); `ops.prepare_gdn_capture()` sizes it on request (the capture
    then works and replays bit-identically), `ops.release_gdn_workspaces()` drops the eager areas and buffers, and the next eager
    call builds its own again."""
    from infinitevl_amd import ops
    run = _gdn_fused_case(4096)
    torch.cuda.synchronize()
    ops.release_gdn_workspaces(DEV)
    dev = torch.device(DEV).index
    assert not [k for k in ops._GDN_SYNC if k[0] == dev]
    ref = run(True)                                                  # eager, single launch, ~64 MB of records
    torch.cuda.synchronize()
    garea = ops._GDN_SYNC[(dev, "graphs")]
    ent = ops._GDN_WS.get(id(garea))
    assert ent is None or ent[0].numel() <= ops._GDN_MIRROR_CAP, "the mirror must not follow a large eager call"
    eager_area = ops._gdn_sync_area(torch.device(DEV))
    assert ops._GDN_WS[id(eager_area)][0].numel() >= 60 << 20
    ops.prepare_gdn_capture(DEV)                                     # (without it the capture below would have to grow the buffer: refused)
    assert ops._GDN_WS[id(garea)][0].numel() >= 60 << 20
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run(True)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    del g, out
    ops.release_gdn_workspaces(DEV)
    assert not [k for k in ops._GDN_SYNC if k[0] == dev] and id(garea) not in ops._GDN_WS
    again = run(True)
    torch.cuda.synchronize()
    assert torch.equal(again[0], ref[0])
