"""configs[4] check: a continuous 1M-token stream (4096 hipGraph steps x 256 tokens, SWA window 4096) through
the 36-layer InfiniteVL-3B text stack on one MI355X.  Reports the step-latency distribution against the 24 FPS
budget (41.7 ms per 256-token frame) and that allocated memory stays flat, with bf16 or e4m3 MFMA operands.

    python tools/stream_1m.py [--steps 4096] [--mma-dtype fp8_e4m3] > gpurun_out/stream_1m.json
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--window", type=int, default=4096)
    ap.add_argument("--mma-dtype", default=None, help="fp8_e4m3 = BASELINE.json configs[4]")
    args = ap.parse_args()
    import bench
    bench._enable_tunableop()
    import infinitevl_amd
    infinitevl_amd.load_library()
    from infinitevl_amd.harness import GraphedStep, InfiniteVLTextConfig, InfiniteVLTextStack
    dev = torch.device("cuda", 0)
    cfg = InfiniteVLTextConfig(sliding_window=args.window)
    with torch.device(dev):
        torch.set_default_dtype(torch.bfloat16)
        model = InfiniteVLTextStack(cfg)
        torch.set_default_dtype(torch.float32)
    model = model.to(torch.bfloat16).eval()
    model.init_weights_(seed=0)
    model.fuse_()
    model.set_mma_dtype(args.mma_dtype)
    cache = model.allocate_inference_cache(1)
    T = args.chunk
    gen = torch.Generator(device=dev).manual_seed(5)
    frames = [(torch.randn(1, T, cfg.hidden_size, device=dev, generator=gen) * 0.02).to(torch.bfloat16) for _ in range(8)]
    step = GraphedStep(model, cache, 1, T, logits_to_keep=1)
    step.capture()
    for i in range(20):
        step.step(frames[i % 8])
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated(dev)
    group, lat = 32, []
    t_all = time.perf_counter()
    for s0 in range(0, args.steps, group):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(s0, min(args.steps, s0 + group)):
            step.step(frames[i % 8])
        e1.record()
        e1.synchronize()
        lat.append(e0.elapsed_time(e1) / (min(args.steps, s0 + group) - s0))
    wall = time.perf_counter() - t_all
    mem1 = torch.cuda.memory_allocated(dev)
    lt = torch.tensor(lat)
    out = {"mma_dtype": args.mma_dtype or "bf16", "steps": args.steps, "tokens": args.steps * T, "context_tokens": cache.get_seq_length(),
           "ms_per_step_mean": float(lt.mean()), "ms_per_step_max_group_of_32": float(lt.max()),
           "ms_per_step_min_group_of_32": float(lt.min()), "frame_budget_ms_24fps": 1000 / 24,
           "fps_equivalent": 1000 / float(lt.mean()), "wall_s": wall,
           "mem_allocated_before": mem0, "mem_allocated_after": mem1, "mem_flat": mem0 == mem1,
           "peak_mem_gib": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
           "logits_finite": bool(torch.isfinite(step.logits.float()).all())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
