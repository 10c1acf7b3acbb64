"""CPU oracle for the InfiniteVL hybrid-attention hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32) / numpy
restatement of the reference's algorithm for the path named in BASELINE.json
(Gated DeltaNet chunk/recurrent rule, short conv, gated RMSNorm, gate math,
sliding-window attention, M-RoPE, and the static cache's integer bookkeeping).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it -- and there only as the checker / reported baseline.  The product
package `infinitevl_amd` never imports it and has no CPU fallback.

Parity pin: every function here is checked in `tests/test_oracle_golden.py`
against fixtures under `tests/golden/` that were produced by executing the
reference's own code in the build container (`tests/golden/gen_golden.py`:
vendored fla Triton kernels under TRITON_INTERPRET=1, the reference's
`eager_attention_forward`, `apply_multimodal_rotary_pos_emb`, cache classes and
layer modules imported from /root/reference).  The third-party pip boundary
(flash-attn 2.7.4.post1, fla-core 0.4.0, causal-conv1d 1.5.0.post5) is absent
from /root/reference and has no reference-side tests: at that boundary parity is
"unpinned" (see DESIGN.md section 3).
"""
from . import gdn, swa, cache, model  # noqa: F401
