import sys; sys.path.insert(0,'.')
import torch
from tools import parity
from oracle import model as omodel
from infinitevl_amd.harness import InfiniteVLTextStack
device="cuda:0"
hc, oc = parity.small_configs(96)
params = parity.bf16_params(omodel.random_params(oc, seed=0, vocab=hc.vocab_size))
stack = InfiniteVLTextStack(hc); parity.load_params(stack, params)
stack = stack.to(device=device, dtype=torch.bfloat16).eval()
cache = stack.allocate_inference_cache(1)
ocache = omodel.new_cache(oc, cache_dtype=torch.bfloat16)
g_ = torch.Generator().manual_seed(1)
pos=0
with torch.no_grad():
    for name,T in [("prefill",130),("stream",70),("d0",1),("d1",1),("d2",1)]:
        x = (torch.randn(1, T, hc.hidden_size, generator=g_) * 0.5).to(torch.bfloat16).float()
        pid = torch.arange(pos, pos + T)[None, None, :].expand(3, 1, T).contiguous()
        omodel.text_stack(params, x, pid, oc, ocache, act_dtype=torch.bfloat16, kernel_rounding=torch.bfloat16)
        stack(inputs_embeds=x.to(device, torch.bfloat16), position_ids=pid.to(device), past_key_values=cache, logits_to_keep=0)
        pos+=T
        k_ref = ocache[0].k[0,0]; k_got = cache.layers[0].keys[0,0].float().cpu()
        v_ref = ocache[0].v[0,0]; v_got = cache.layers[0].values[0,0].float().cpu()
        err = (k_ref-k_got).norm(dim=-1)/k_ref.norm(dim=-1)
        errv = (v_ref-v_got).norm(dim=-1)/v_ref.norm(dim=-1)
        print(name, "size", cache.layers[0].size, "k rows err max %.3e mean %.3e argmax %d | v max %.3e"%(err.max(), err.mean(), int(err.argmax()), errv.max()))
        if name=="prefill":
            print(" per-channel err:", ((k_ref-k_got).abs().mean(0)/k_ref.abs().mean(0))[:8], ((k_ref-k_got).abs().mean(0)/k_ref.abs().mean(0))[60:68])
