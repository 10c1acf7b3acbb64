"""Sequence-parallel prefill check on ONE GPU: two processes (gloo; RCCL refuses two ranks on one device) share
cuda:0, each runs the real HIP layer stack on its segment of one sequence with the per-layer state hand-off of
infinitevl_amd.dist.sequence_parallel_prefill; the last rank compares against the same two calls made back to back in
a single process: hidden states, every cache tensor and two decode steps must match bit for bit.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 tools/sp_check.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from infinitevl_amd import dist as ivd
    from infinitevl_amd.harness import InfiniteVLTextConfig, InfiniteVLTextStack
    rank, world, _ = ivd.init_distributed("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    heads, hidden, window = 2, 256, 96
    lt = ["sliding_attention" if i % 4 == 0 else "linear_attention" for i in range(8)]      # two sliding layers: shared counter
    cfg = InfiniteVLTextConfig(vocab_size=512, hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=8,
                               num_attention_heads=heads, num_key_value_heads=1, head_dim=128, sliding_window=window,
                               layer_types=lt, num_linear_heads=heads, num_linear_key_value_heads=heads,
                               linear_head_dim=128, rope_theta=1e6)
    model = InfiniteVLTextStack(cfg).to(device=dev, dtype=torch.bfloat16).eval()
    model.init_weights_(seed=7)
    model.fuse_()
    total, B = 200, 1
    g_ = torch.Generator().manual_seed(11)
    xs = (torch.randn(B, total, hidden, generator=g_) * 0.5).to(torch.bfloat16).to(dev)
    dec = [(torch.randn(B, 1, hidden, generator=g_) * 0.5).to(torch.bfloat16).to(dev) for _ in range(2)]
    first, last = ivd.segment_bounds(total, rank, world)               # 128 | 72
    cache = model.allocate_inference_cache(B)
    with torch.no_grad():
        h, _ = ivd.sequence_parallel_prefill(model, xs[:, first:last], cache, first)
    torch.cuda.synchronize()
    ok = True
    if rank == world - 1:
        with torch.no_grad():
            outs = [model(inputs_embeds=d_, past_key_values=cache, logits_to_keep=1) for d_ in dec]
            ref_cache = model.allocate_inference_cache(B)
            b0 = ivd.segment_bounds(total, 0, world)
            model(inputs_embeds=xs[:, b0[0]:b0[1]], past_key_values=ref_cache, logits_to_keep=0)
            h_ref, _ = model(inputs_embeds=xs[:, first:last], past_key_values=ref_cache, logits_to_keep=0)
            ref_outs = [model(inputs_embeds=d_, past_key_values=ref_cache, logits_to_keep=1) for d_ in dec]
        torch.cuda.synchronize()
        checks = {"segment hidden": torch.equal(h, h_ref)}
        for i, ((ha, la), (hb, lb)) in enumerate(zip(outs, ref_outs)):
            checks[f"decode {i}"] = torch.equal(ha, hb) and torch.equal(la, lb)
        for i, (la, lb) in enumerate(zip(cache.layers, ref_cache.layers)):
            checks[f"cache layer {i}"] = all(torch.equal(ta, tb) for ta, tb in
                                            zip(la.carried_tensors(), lb.carried_tensors()))
        checks["seq_length"] = cache.get_seq_length() == ref_cache.get_seq_length() == total + 2
        ok = all(checks.values())
        if not ok:
            print("SP_CHECK details:", checks, cache.get_seq_length(), ref_cache.get_seq_length(), flush=True)
        if len(sys.argv) > 2 and sys.argv[1] == "--dump":        # for the test's oracle comparison (tests/ may use oracle/, tools/ may not)
            torch.save({"state_dict": {k: v.detach().float().cpu() for k, v in model.state_dict().items()}, "xs": xs.float().cpu(),
                        "h_last_segment": h.float().cpu(), "first": first, "last": last, "window": window, "layers": len(lt), "heads": heads},
                       sys.argv[2])
        print("SP_CHECK", "PASS" if ok else "FAIL", flush=True)
    ivd.barrier()
    torch.distributed.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
