import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from infinitevl_amd import ops, _lib
if os.environ.get("IVL_LIB"):
    _lib.load(os.environ["IVL_LIB"])
ops._PRENORM = os.environ["IVL_PRENORM"] == "1"
from infinitevl_amd.harness import GraphedDecode, InfiniteVLTextConfig, InfiniteVLTextStack
dev = torch.device("cuda", 0)
cfg = InfiniteVLTextConfig()
stack = InfiniteVLTextStack(cfg).to(dev).to(torch.bfloat16).init_weights_(seed=0).fuse_()
c = stack.allocate_inference_cache(1)
x = (torch.randn(1, 256, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(20): stack(inputs_embeds=x, past_key_values=c, logits_to_keep=1)
gd = GraphedDecode(stack, c, 1)
for _ in range(5): gd.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(64): gd.step()
e1.record(); torch.cuda.synchronize()
print("PRENORM", ops._PRENORM, "ms/token", e0.elapsed_time(e1) / 64)
