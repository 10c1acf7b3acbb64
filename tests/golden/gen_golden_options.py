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


if __name__ == "__main__":
    main()
