#!/usr/bin/env python3
"""Developer A/B inside one process: ms per 256-token streaming step (captured graph, full-width 36-layer stack, full window) with the
single-launch GDN call on / off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinitevl_amd import ops
from infinitevl_amd.harness import GraphedStep, InfiniteVLTextConfig, InfiniteVLTextStack
dev = torch.device("cuda", 0)
cfg = InfiniteVLTextConfig()
stack = InfiniteVLTextStack(cfg).to(dev).to(torch.bfloat16).init_weights_(seed=0).fuse_()
x = (torch.randn(1, 256, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
res = {}
for name, flag in (("two launches", False), ("single launch", True), ("two launches", False), ("single launch", True)):
    ops._GDN_SINGLE_LAUNCH = flag
    c = stack.allocate_inference_cache(1)
    with torch.no_grad():
        for _ in range(20):
            stack(inputs_embeds=x, past_key_values=c, logits_to_keep=1)       # fill the window
    gs = GraphedStep(stack, c, 1, 256, logits_to_keep=1)
    gs.inputs_embeds.copy_(x)
    for _ in range(5):
        gs.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(64):
        gs.step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 64
    res.setdefault(name, []).append(ms)
    print(f"GDN fused call as {name}: {ms:.4f} ms per 256-token step")
ops._GDN_SINGLE_LAUNCH = True
