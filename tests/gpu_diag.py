#!/usr/bin/env python3
"""GPU bring-up diagnostics: runs every kernel through the C ABI against the oracle, never stops at
the first failure, prints a table and writes gpurun_out/diag.json.  Not a test (tests/ has those);
this is the tool used while bringing kernels up over `gpurun` round trips.

    python tools/gpu_diag.py [--only substr] [--quick]
"""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import gdn as ogdn  # noqa: E402
from oracle import swa as oswa  # noqa: E402
import parity  # noqa: E402  (tests/ is this script's directory)

RESULTS = {}


def check(name):
    def deco(fn):
        fn._check_name = name
        CHECKS.append(fn)
        return fn
    return deco


CHECKS = []
DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16)


@check("short_conv")
def _conv():
    from infinitevl_amd import ops
    out = {}
    torch.manual_seed(0)
    for (B, T, D) in [(2, 10, 64), (1, 1, 2048), (1, 257, 4096), (2, 3, 64), (1, 70, 512)]:
        w = bf(torch.randn(D, 1, 4) * 0.5)
        x = bf(torch.randn(B, T, D))
        st = bf(torch.randn(B, D, 4))
        conv = ops.ShortConvolution(D, 4).to(DEV, torch.bfloat16)
        with torch.no_grad():
            conv.weight.copy_(w)
        for label, state in (("nostate", None), ("state", st)):
            y_ref, s_ref = ogdn.short_conv(x.float(), w.float().reshape(D, 4), None if state is None else state.float())
            cache = None if state is None else state.clone().to(DEV)
            y, c = conv(x.to(DEV), cache=cache, output_final_state=True)
            torch.cuda.synchronize()
            out[f"B{B}T{T}D{D}_{label}_y"] = parity.rms_rel(y_ref, y.float())
            out[f"B{B}T{T}D{D}_{label}_state_exact"] = float(torch.equal(s_ref.to(torch.bfloat16), c.cpu()))
    return out


@check("rmsnorm_gate")
def _norm():
    from infinitevl_amd import ops
    torch.manual_seed(1)
    x = bf(torch.randn(3, 37, 16, 256) * 2)
    g = bf(torch.randn(3, 37, 16, 256))
    n = ops.FusedRMSNormGated(256, eps=1e-5).to(DEV, torch.bfloat16)
    w = bf(1 + 0.1 * torch.randn(256))
    with torch.no_grad():
        n.weight.copy_(w)
    y = n(x.to(DEV), g.to(DEV))
    ref = ogdn.rmsnorm_swish_gate(x.float(), g.float(), w.float(), 1e-5)
    return {"y": parity.rms_rel(ref, y.float())}


@check("gate_math")
def _gate():
    from infinitevl_amd import ops
    torch.manual_seed(2)
    a = bf(torch.randn(2, 50, 16) * 3)
    b = bf(torch.randn(2, 50, 16) * 3)
    A_log = torch.log(torch.empty(16).uniform_(0.5, 16))
    dt = torch.randn(16) * 0.5
    g, beta = ops.gdn_gate(a.to(DEV), b.to(DEV), A_log.to(DEV), dt.to(DEV))
    g_ref, b_ref = ogdn.gate_math(a, b, A_log, dt)
    return {"g": parity.rms_rel(g_ref, g), "beta_exact": float(torch.equal(b_ref, beta.cpu()))}


@check("mrope")
def _mrope():
    from infinitevl_amd import ops
    torch.manual_seed(3)
    B, T, Hq, Hkv, d = 2, 33, 16, 2, 128
    q = bf(torch.randn(B, T, Hq, d))
    k = bf(torch.randn(B, T, Hkv, d))
    pos = torch.stack([torch.arange(1000, 1000 + T), torch.randint(0, 60, (T,)), torch.randint(0, 60, (T,))])
    pos = pos[:, None, :].expand(3, B, T).contiguous()
    cos, sin = oswa.rotary_cos_sin(pos, d, 1e6)
    cos, sin = bf(cos), bf(sin)
    # reference arithmetic in bf16 eager (std:974-984)
    qe, ke = oswa.apply_mrope(q.transpose(1, 2), k.transpose(1, 2), cos, sin, [16, 24, 24])
    qd, kd = q.clone().to(DEV), k.clone().to(DEV)
    ops.apply_mrope_inplace(qd, kd, cos.to(DEV), sin.to(DEV), [16, 24, 24])
    torch.cuda.synchronize()
    return {"q_bit_exact": float(torch.equal(qe.transpose(1, 2).contiguous(), qd.cpu())),
            "k_bit_exact": float(torch.equal(ke.transpose(1, 2).contiguous(), kd.cpu())),
            "q_rms": parity.rms_rel(qe.transpose(1, 2).float(), qd.float())}


@check("gdn_recurrent")
def _rec():
    out = {}
    for (B, T, H, h0, sd, inplace) in [(1, 1, 2, True, torch.float32, False), (2, 7, 2, True, torch.float32, False),
                                       (1, 64, 2, False, torch.float32, False), (1, 33, 16, True, torch.bfloat16, True),
                                       (1, 70, 2, True, torch.float32, False)]:
        r = parity.gdn_op_parity(DEV, "recurrent", B, T, H, seed=T, with_h0=h0, state_dtype=sd, inplace_state=inplace)
        for k, v in r.items():
            out[f"B{B}T{T}H{H}_{k}"] = v
    return out


@check("gdn_chunk")
def _chunk():
    out = {}
    for (B, T, H, h0, sd, inplace) in [(1, 64, 1, False, torch.float32, False), (1, 65, 2, True, torch.float32, False),
                                       (1, 160, 2, False, torch.float32, False), (2, 256, 2, True, torch.float32, False),
                                       (1, 256, 16, True, torch.bfloat16, True), (1, 1, 2, True, torch.float32, False),
                                       (1, 30, 2, True, torch.float32, False)]:
        r = parity.gdn_op_parity(DEV, "chunk", B, T, H, seed=T + 1, with_h0=h0, state_dtype=sd, inplace_state=inplace)
        for k, v in r.items():
            out[f"B{B}T{T}H{H}_{k}"] = v
    return out


@check("gdn_chunk_golden")
def _chunk_golden():
    """Committed fixtures produced by the reference's own kernels."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    from infinitevl_amd import ops
    out = {}
    for name in ["rec_T1_h0", "rec_T7_h0", "rec_T64", "chunk_T65_h0", "chunk_T160", "chunk_T256_h0"]:
        z = load_golden("gdn_" + name)
        fn = ops.fused_recurrent_gated_delta_rule if name.startswith("rec") else ops.chunk_gated_delta_rule
        h0 = z.get("h0")
        o, ht = fn(bf(z["q"]).to(DEV), bf(z["k"]).to(DEV), bf(z["v"]).to(DEV), z["g"].to(DEV), bf(z["beta"]).to(DEV),
                   initial_state=None if h0 is None else h0.to(DEV), output_final_state=True,
                   use_qk_l2norm_in_kernel=True)
        out[name + "_o"] = parity.rms_rel(z["o"], o.float())
        out[name + "_ht"] = parity.rms_rel(z["ht"], ht.float())
    return out


@check("swa")
def _swa():
    out = {}
    cases = [
        # B, T, Hq, Hkv, W, seen, via
        (1, 5, 2, 1, 8, 0, "cat"), (2, 19, 4, 2, 8, 0, "cat"), (1, 6, 16, 2, 8, 9, "cat"),
        (1, 64, 2, 1, 4096, 0, "cat"), (1, 130, 16, 2, 4096, 0, "cat"), (1, 200, 2, 1, 96, 0, "cat"),
        (1, 70, 2, 1, 96, 250, "ring"), (1, 256, 16, 2, 4096, 4500, "ring"), (1, 256, 16, 2, 4096, 1000, "ring"),
        (1, 1, 16, 2, 4096, 5000, "ring"), (2, 1, 16, 2, 96, 40, "ring"), (1, 3, 16, 2, 96, 500, "ring"),
        (1, 300, 2, 1, 96, 77, "ring"),
    ]
    for (B, T, Hq, Hkv, W, seen, via) in cases:
        r = parity.swa_op_parity(DEV, B, T, Hq, Hkv, W, seen, seed=T + seen, via=via)
        for k, v in r.items():
            out[f"B{B}T{T}Hq{Hq}W{W}seen{seen}_{via}_{k}"] = v
    return out


@check("layers")
def _layers():
    return parity.layer_parity(DEV, T_prefill=130, n_decode=3, window=96, seed=0)


@check("graph_step")
def _graph():
    """hipGraph-captured streaming step == eager step (state + outputs), replayed several times."""
    from infinitevl_amd.harness import GraphedStep, InfiniteVLTextStack
    from oracle import model as omodel
    hc, oc = parity.small_configs(window=96)
    params = parity.bf16_params(omodel.random_params(oc, seed=3, vocab=hc.vocab_size))
    stack = InfiniteVLTextStack(hc)
    parity.load_params(stack, params)
    stack = stack.to(DEV, torch.bfloat16).eval()
    T = 70
    xs = [bf(torch.randn(1, T, hc.hidden_size) * 0.5).to(DEV) for _ in range(5)]
    with torch.no_grad():
        c1 = stack.allocate_inference_cache(1)
        eager = []
        pos = 0
        for x in xs:
            pid = torch.arange(pos, pos + T, device=DEV)[None, None, :].expand(3, 1, T)
            h, _ = stack(inputs_embeds=x, position_ids=pid, past_key_values=c1, logits_to_keep=0)
            eager.append(h.clone())
            pos += T
        c2 = stack.allocate_inference_cache(1)
        # first frame eager (demo:326-347), then graph
        pid = torch.arange(0, T, device=DEV)[None, None, :].expand(3, 1, T)
        h0, _ = stack(inputs_embeds=xs[0], position_ids=pid, past_key_values=c2, logits_to_keep=0)
        gs = GraphedStep(stack, c2, 1, T, logits_to_keep=0)
        out = {"frame0": parity.rms_rel(eager[0].float(), h0.float())}
        for i in range(1, 5):
            h, _ = gs.step(xs[i])
            torch.cuda.synchronize()
            out[f"frame{i}"] = parity.rms_rel(eager[i].float(), h.float())
            out[f"frame{i}_bit_exact"] = float(torch.equal(eager[i], h))
        out["host_counters_match"] = float(c1.layers[0].cumulative_length == c2.layers[0].cumulative_length
                                           and c1.layers[0].size == c2.layers[0].size
                                           and c1.layers[1].seq_len == c2.layers[1].seq_len)
        out["dev_counter_match"] = float(int(c2.layers[0]._pos_dev.item()) == c2.layers[0].cumulative_length)
        out["state_bit_exact"] = float(torch.equal(c1.layers[1].recurrent_state, c2.layers[1].recurrent_state))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "diag.json"))
    args = ap.parse_args()
    import infinitevl_amd
    infinitevl_amd.load_library()
    print("device:", torch.cuda.get_device_name(0))
    for fn in CHECKS:
        name = fn._check_name
        if args.only and args.only not in name:
            continue
        t0 = time.time()
        try:
            RESULTS[name] = fn()
            status = "ok"
        except Exception as e:  # noqa: BLE001
            RESULTS[name] = {"EXCEPTION": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-1500:]}
            status = "EXC"
        print(f"=== {name} [{status}] {time.time() - t0:.1f}s")
        for k, v in RESULTS[name].items():
            print(f"    {k:56s} {v if isinstance(v, str) else format(v, '.3e')}")
        sys.stdout.flush()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(RESULTS, f, indent=1)


if __name__ == "__main__":
    main()
