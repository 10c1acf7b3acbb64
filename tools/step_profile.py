#!/usr/bin/env python3
"""Experiment: per-kernel durations INSIDE the real streaming step, measured live with torch.profiler (roctracer /
rocprofiler-sdk activity records of hipGraph replays) instead of an external rocprofv3 run.
usage: step_profile.py [steps=4] [layers=36]"""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (enables TunableOp before torch is imported)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 36
    import infinitevl_amd
    infinitevl_amd.load_library()
    from infinitevl_amd.harness import GraphedStep, InfiniteVLTextConfig, InfiniteVLTextStack
    dev = torch.device("cuda", 0)
    cfg = InfiniteVLTextConfig(sliding_window=4096, num_hidden_layers=layers)
    with torch.device(dev):
        torch.set_default_dtype(torch.bfloat16)
        model = InfiniteVLTextStack(cfg)
        torch.set_default_dtype(torch.float32)
    model = model.to(torch.bfloat16).eval()
    model.init_weights_(seed=0)
    model.fuse_()
    cache = model.allocate_inference_cache(1)
    T = 256
    x = (torch.randn(1, T, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
    step = GraphedStep(model, cache, 1, T, logits_to_keep=1)
    step.capture()
    for _ in range(20):
        step.step(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(steps):
            step.step(x)
        torch.cuda.synchronize()
    agg = collections.defaultdict(list)
    for ev in prof.events():
        if ev.device_type is not None and "cuda" in str(ev.device_type).lower():
            agg[ev.name.split("(")[0]].append(ev.device_time if hasattr(ev, "device_time") else ev.cuda_time)
    rows = sorted(((n, len(v), sum(v) / len(v), sum(v)) for n, v in agg.items()), key=lambda r: -r[3])
    out = [{"kernel": n[:120], "launches": c, "avg_us": a, "total_us": t} for n, c, a, t in rows[:40]]
    print(json.dumps({"steps": steps, "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
