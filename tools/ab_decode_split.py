#!/usr/bin/env python3
"""A/B of the decode token through the full-size stack: the split GDN decode step (64 workgroups + norm / conv-state shift in the
o_proj launch) against the one-launch step + plain o_proj (ops._SPLIT_DECODE), graphed greedy decode, ms per token, ABAB."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinitevl_amd import ops
from infinitevl_amd.harness import GraphedDecode, InfiniteVLTextConfig, InfiniteVLTextStack
dev = torch.device("cuda", 0)
cfg = InfiniteVLTextConfig(sliding_window=4096)
with torch.device(dev):
    torch.set_default_dtype(torch.bfloat16)
    model = InfiniteVLTextStack(cfg)
    torch.set_default_dtype(torch.float32)
model = model.to(torch.bfloat16).eval()
model.init_weights_(seed=0).fuse_()
x = (torch.randn(1, 4096, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
decs = {}
with torch.no_grad():
    for flag in (True, False):
        ops._SPLIT_DECODE = flag
        cache = model.allocate_inference_cache(1)
        model(inputs_embeds=x, past_key_values=cache)
        model(inputs_embeds=x, past_key_values=cache)          # 8192 tokens: full window
        d = GraphedDecode(model, cache, 1)
        d.capture()
        decs[flag] = d
    ops._SPLIT_DECODE = False
    for rep in range(3):
        for flag in (True, False):
            d = decs[flag]
            for _ in range(8):
                d.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 128
            for _ in range(n):
                d.step()
            torch.cuda.synchronize()
            print(f"split={flag}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per token", flush=True)
