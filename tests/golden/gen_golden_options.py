#!/usr/bin/env python3
"""Golden fixtures for the NON-DEFAULT options of the fla modules (round 4), produced like gen_golden.py by executing the
reference in the build container (vendored Triton kernels under TRITON_INTERPRET=1, shims on our side: _ref_loader.py):

    short_conv_bias.npz      ShortConvolution(bias=True)                        fla:modules/convolution.py:128-293
    rmsnorm_gate_options.npz FusedRMSNormGated(elementwise_affine=False), residual=, prenorm=, residual_in_fp32=
                                                                              fla:modules/fused_norm_gate.py:735-796

    python tests/golden/gen_golden_options.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")
from _ref_loader import load_reference_fla  # noqa: E402
from gen_golden import bits, save, snap  # noqa: E402

import torch  # noqa: E402


def main():
    ns = load_reference_fla()
    torch.manual_seed(4)
    D, W = 64, 4
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        conv = ns.conv.ShortConvolution(D, W, bias=True, activation="silu", use_fast_conv1d=False)
    with torch.no_grad():
        conv.weight.copy_(snap(torch.randn(D, 1, W) * 0.5))
        conv.bias.copy_(snap(torch.randn(D) * 0.5))
    x = snap(torch.randn(2, 11, D))
    with torch.no_grad():
        y, cache = conv(x, cache=None, output_final_state=True)
        xs = snap(torch.randn(3, 2, 1, D))
        ys, caches, c = [], [], cache.clone()
        for i in range(3):
            yi, c = conv(xs[i], cache=c, output_final_state=True)
            ys.append(yi.clone())
            caches.append(c.clone())
    save("short_conv_bias", weight_bf16bits=bits(conv.weight.detach()), bias_bf16bits=bits(conv.bias.detach()), x_bf16bits=bits(x),
         y=y, state=cache, xs_bf16bits=bits(xs), ys=torch.stack(ys), states=torch.stack(caches))

    torch.manual_seed(5)
    xo = snap(torch.randn(2, 5, 4, 256) * 2.0)
    gate = snap(torch.randn(2, 5, 4, 256))
    res = snap(torch.randn(2, 5, 4, 256))
    plain = ns.norm_gate.FusedRMSNormGated(256, elementwise_affine=False, eps=1e-5)
    aff = ns.norm_gate.FusedRMSNormGated(256, eps=1e-5)
    with torch.no_grad():
        aff.weight.copy_(snap(1.0 + 0.1 * torch.randn(256)))
        y_plain = plain(xo, gate)
        y_res = aff(xo, gate, residual=res)
        y_pre, r_pre = aff(xo, gate, residual=res, prenorm=True)
        y_pre32, r_pre32 = aff(xo, gate, prenorm=True, residual_in_fp32=True)
    save("rmsnorm_gate_options", weight_bf16bits=bits(aff.weight.detach()), x_bf16bits=bits(xo), gate_bf16bits=bits(gate),
         residual_bf16bits=bits(res), y_no_affine=y_plain, y_residual=y_res, y_prenorm=y_pre, residual_out=r_pre,
         y_prenorm_fp32=y_pre32, residual_out_fp32=r_pre32, eps=np.float32(1e-5))


def gen_varlen(ns):
    """cu_seqlens inputs of the two GDN operators (fla:ops/gated_delta_rule/chunk.py:355-369, fused_recurrent.py:296-312):
    four sequences of 70 / 100 / 1 / 79 tokens flattened into one [1, 250, ...] batch, one initial state each."""
    from gen_golden import gdn_inputs
    cu = torch.tensor([0, 70, 170, 171, 250], dtype=torch.long)
    q, k, v, g, beta, _ = gdn_inputs(7, 1, 250, 2, 128, 256, False)
    h0 = snap(torch.randn(4, 2, 128, 256, generator=torch.Generator().manual_seed(8)))
    scale = 128 ** -0.5
    qn, kn = ns.l2norm.l2norm_fwd(q.contiguous()), ns.l2norm.l2norm_fwd(k.contiguous())
    import triton
    idx = torch.cat([torch.arange(n) for n in triton.cdiv(cu[1:] - cu[:-1], 64).tolist()])          # chunk.py:207-214
    idx = torch.stack([idx.eq(0).cumsum(0) - 1, idx], 1).to(cu)
    out = ns.chunk.chunk_gated_delta_rule_fwd(qn, kn, v.contiguous(), g.contiguous(), beta.contiguous(), scale, h0, True,
                                              offsets=cu, indices=idx, head_first=False)
    o_c, ht_c = out[1], out[-1]
    o_r, ht_r = ns.recurrent.fused_recurrent_gated_delta_rule(q, k, v, g, beta, scale=scale, initial_state=h0, output_final_state=True,
                                                              cu_seqlens=cu, head_first=False, use_qk_l2norm_in_kernel=True)
    save("gdn_varlen", q_bf16bits=bits(q), k_bf16bits=bits(k), v_bf16bits=bits(v), beta_bf16bits=bits(beta), g=g,
         h0_bf16bits=bits(h0), cu_seqlens=cu.numpy(), o_chunk=o_c.float(), ht_chunk=ht_c.float(), o_recurrent=o_r.float(),
         ht_recurrent=ht_r.float())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "varlen":
        gen_varlen(load_reference_fla())
    else:
        main()
        gen_varlen(load_reference_fla())
