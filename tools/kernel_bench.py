#!/usr/bin/env python3
"""Micro-benchmark of the hot-path kernels at the bench shapes (same code as bench.py's `kernels`
leg); used for A/B timing and as the small command profiled with rocprofv3 --pmc.

    python tools/kernel_bench.py [--chunk 256] [--window 4096] [--only substr] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--window", type=int, default=4096)
    ap.add_argument("--json", default="")
    ap.add_argument("--only", default=None)
    ap.add_argument("--lib", default=None, help="A/B timing against another build of the library (developer use)")
    args = ap.parse_args()
    import infinitevl_amd
    if args.lib:
        from infinitevl_amd import _lib
        _lib.load(args.lib)
    infinitevl_amd.load_library()
    dev = torch.device("cuda", 0)
    res = bench.kernel_timings(dev, args.chunk, args.window, only=args.only)
    for k, v in res.items():
        print(f"{k:34s} {v['ms'] * 1e3:9.2f} us/launch  x{v['launches_per_step']:3d}/step  "
              f"{v['achieved']:9.1f} {v['unit']:8s} frac={v['frac']:.4f}")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
