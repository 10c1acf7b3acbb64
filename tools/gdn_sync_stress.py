#!/usr/bin/env python3
"""Stress of the single-launch forms of the fused GDN call (ivl_gdn_chunk_fused_fwd with a sync area) against the two-launch
form, bit for bit, under contention.  Started N times at once by tests/test_gpu_parity.py (N processes sharing one MI355X:
each has its own sync area, all of them compete for the CUs), or alone with --co-stream (a second stream of this process keeps
the chip full of streaming kernels).

usage: gdn_sync_stress.py [--T 1000,4300] [--H 16] [--B 1] [--iters 6] [--co-stream] [--barrier-file PATH --nprocs N]
prints one line `GDN_SYNC_STRESS PASS ...` or `... FAIL ...` and exits 0 / 1.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from infinitevl_amd import ops  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", default="1000,4300")
    ap.add_argument("--H", type=int, default=16)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--sets", type=int, default=3)
    ap.add_argument("--co-stream", action="store_true")
    ap.add_argument("--barrier-file", default=None)
    ap.add_argument("--nprocs", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, H, K, V = a.B, a.H, 128, 256
    Dq, Dk, Dv = H * K, H * K, H * V
    g_ = torch.Generator(device=dev).manual_seed(1234 + a.seed)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g_).to(torch.bfloat16)  # noqa: E731
    cols = (0, Dq, Dq + Dk, Dq + Dk + 2 * Dv, Dq + Dk + 2 * Dv + H)
    ld = cols[4] + H
    cw = [rn(D_, 1, 4) * 0.5 for D_ in (Dq, Dk, Dv)]
    A32, dt32 = torch.randn(H, device=dev, generator=g_), torch.randn(H, device=dev, generator=g_)

    def run(st, single):
        proj, cs, h0 = st
        ops._GDN_SINGLE_LAUNCH = single
        try:
            so = [c.clone() for c in cs]
            ht = torch.zeros_like(h0)
            o = ops.gdn_chunk_fused(proj, cols, cw, so, so, A32, dt32, H, K, V, initial_state=h0, final_state_out=ht)
        finally:
            ops._GDN_SINGLE_LAUNCH = True
        return o, ht, so

    cases = []
    for T in [int(x) for x in a.T.split(",")]:
        sets = [(rn(B, T, ld), [rn(B, D_, 4) for D_ in (Dq, Dk, Dv)],
                 (torch.randn(B, H, K, V, device=dev, generator=g_) * 0.1).to(torch.bfloat16)) for _ in range(a.sets)]
        refs = [run(st, False) for st in sets]
        cases.append((T, sets, refs))
    torch.cuda.synchronize()

    # all processes start their single-launch phase together
    if a.barrier_file:
        with open(f"{a.barrier_file}.{os.getpid()}", "w") as f:
            f.write("ready")
        d, base = os.path.dirname(a.barrier_file), os.path.basename(a.barrier_file)
        t0 = time.time()
        while len([x for x in os.listdir(d) if x.startswith(base + ".")]) < a.nprocs and time.time() - t0 < 120:
            time.sleep(0.01)

    co = None
    if a.co_stream:
        # a chip-filling streaming kernel on a second stream for the whole single-launch phase: 2 GiB copies, back to back
        co = torch.cuda.Stream()
        src = torch.empty(1 << 29, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        small = torch.empty(1 << 22, dtype=torch.float32, device=dev)

    bad, calls = 0, 0
    for it in range(a.iters):
        if co is not None:
            with torch.cuda.stream(co):
                for _ in range(4):
                    dst.copy_(src)
                    small.mul_(1.0001)          # many short workgroups between the long copies: uneven load
        for T, sets, refs in cases:
            outs = [run(st, True) for st in sets]
            torch.cuda.synchronize(dev) if co is None else torch.cuda.current_stream().synchronize()
            for (o2, ht2, so2), (o1, ht1, so1) in zip(refs, outs):
                calls += 1
                ok = torch.equal(o1, o2) and torch.equal(ht1, ht2) and all(torch.equal(x, y) for x, y in zip(so1, so2))
                bad += 0 if ok else 1
    torch.cuda.synchronize()
    err = ""
    try:
        ops.gdn_sync_check(dev, deep=True)
    except Exception as e:  # noqa: BLE001
        err = f" sync error: {e}"
    leftover = sum(int(v.view(torch.int32).abs().sum()) for v in ops._all_gdn_sync_areas(0))
    ok = bad == 0 and not err and leftover == 0
    print(f"GDN_SYNC_STRESS {'PASS' if ok else 'FAIL'} pid={os.getpid()} calls={calls} mismatches={bad} flags_left={leftover}{err}",
          flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
