#!/usr/bin/env python3
"""Developer A/B: ms per 4096-token prefill call over a full ring (the cfg3 leg's call) with the library given as argv[1]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinitevl_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib.load(sys.argv[1])
from infinitevl_amd.harness import InfiniteVLTextConfig, InfiniteVLTextStack
dev = torch.device("cuda", 0)
cfg = InfiniteVLTextConfig()
stack = InfiniteVLTextStack(cfg).to(dev).to(torch.bfloat16).init_weights_(seed=0).fuse_()
c = stack.allocate_inference_cache(1)
x = (torch.randn(1, 4096, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
ts = []
with torch.no_grad():
    for r in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        stack(inputs_embeds=x, past_key_values=c, logits_to_keep=1)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'default'}: ms per 4096-token call: mean {sum(ts[3:]) / len(ts[3:]) * 1e3:.3f} min {min(ts[3:]) * 1e3:.3f}")
