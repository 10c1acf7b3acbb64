#!/usr/bin/env python3
"""Developer check of a build of the 256-row attention path (optionally another library: argv[1]): the 4096-token call over a full ring on
the 256-row kernel vs the 128-row kernel of the same library -- ring bit-equal, output difference -- and a smaller odd-capacity case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from infinitevl_amd import _lib
if len(sys.argv) > 1:
    _lib.load(sys.argv[1])
from infinitevl_amd import ops
import test_gpu_ring256 as t

for (B, T, W, seen) in ((1, 4096, 4096, 9000), (4, 1024, 1000, 5000), (2, 2048, 700, 699)):
    q, k_all, v_all = t._inputs(B, T, 16, 2, seen, seed=5)
    cos, sin = t._rope_tables(B, T, seen)
    outs, rings = [], []
    for pm in (seen, 0):
        kc, vc, pos_dev = t._filled_ring(k_all, v_all, seen, W)
        o = ops.swa_forward(q.to("cuda:0"), k_all[:, seen:].to("cuda:0"), v_all[:, seen:].to("cuda:0"), window=W, scaling=128 ** -0.5, k_cache=kc,
                            v_cache=vc, pos_dev=pos_dev, append=True, pos_min=pm, rope=(cos, sin, (16, 24, 24)))
        torch.cuda.synchronize()
        outs.append(o.float()); rings.append((kc, vc))
    e = float((outs[0] - outs[1]).norm() / outs[1].norm())
    print(f"B={B} T={T} W={W}: ring equal {torch.equal(rings[0][0], rings[1][0]) and torch.equal(rings[0][1], rings[1][1])}, "
          f"rel diff {e:.3e}, finite {bool(torch.isfinite(outs[0]).all())}")
