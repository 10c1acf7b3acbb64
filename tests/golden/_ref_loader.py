"""Loader for the READ-ONLY reference at /root/reference (this container only).

Used exclusively by tests/golden/gen_golden.py to produce the committed fixture
files.  Nothing here (and nothing under /root/reference) is needed at test time:
the fixtures are plain data.

What it does (recipe from SURVEY.md section 8c):
  * runs the vendored flash-linear-attention Triton kernels under the Triton CPU
    interpreter (TRITON_INTERPRET=1) -- there is no GPU in the build container;
  * registers empty `fla`, `fla.ops`, `fla.modules` package shells whose __path__
    points into /root/reference/src/llamafactory/model/fla so that submodules can
    be imported without executing fla/__init__.py (which imports every model and a
    hard-coded /running_package path);
  * replaces triton.autotune by a "first config" wrapper (the autotuner needs a GPU
    driver to benchmark);
  * patches fla.utils.custom_device_ctx to a nullcontext (torch.cpu.device does not
    exist).
No reference source is copied; the modules are imported from where they lie.
"""
import contextlib
import importlib
import os
import sys
import types

REF_ROOT = "/root/reference"
FLA_DIR = os.path.join(REF_ROOT, "src/llamafactory/model/fla")


def _first_config_autotune(configs=None, key=None, **_kw):
    """Stand-in for triton.autotune: always launch the first config."""

    def deco(fn):
        cfg = configs[0] if configs else None

        class _Wrapper:
            def __init__(self, fn):
                self.fn = fn
                self.arg_names = getattr(fn, "arg_names", [])
                self.__name__ = getattr(fn, "__name__", "kernel")

            def run(self, *args, grid=None, warmup=False, **kwargs):
                extra = dict(cfg.kwargs) if cfg is not None else {}
                extra.update(kwargs)
                extra.pop("num_warps", None)
                extra.pop("num_stages", None)
                return self.fn.run(*args, grid=grid, warmup=warmup, **extra)

            def __getitem__(self, grid):
                def launcher(*args, **kwargs):
                    return self.run(*args, grid=grid, warmup=False, **kwargs)

                return launcher

        return _Wrapper(fn)

    return deco


def load_reference_fla():
    """Return a namespace with the vendored reference ops, interpreter-backed."""
    if not os.path.isdir(FLA_DIR):
        raise RuntimeError(f"{FLA_DIR} not found: fixtures can only be (re)generated in the build container")
    os.environ["TRITON_INTERPRET"] = "1"
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    import triton  # noqa: E402  (after TRITON_INTERPRET)

    triton.autotune = _first_config_autotune

    for name, sub in (("fla", ""), ("fla.ops", "ops"), ("fla.modules", "modules")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(FLA_DIR, sub) if sub else FLA_DIR]
            m.__package__ = name
            sys.modules[name] = m

    futils = importlib.import_module("fla.utils")
    futils.custom_device_ctx = lambda index: contextlib.nullcontext()

    ns = types.SimpleNamespace()
    ns.l2norm = importlib.import_module("fla.modules.l2norm")
    ns.cumsum = importlib.import_module("fla.ops.utils.cumsum")
    ns.chunk = importlib.import_module("fla.ops.gated_delta_rule.chunk")
    ns.recurrent = importlib.import_module("fla.ops.gated_delta_rule.fused_recurrent")
    ns.wy = importlib.import_module("fla.ops.gated_delta_rule.wy_fast")
    ns.conv = importlib.import_module("fla.modules.convolution")
    # fla's 'silu'/'swish' activations are torch.cuda.jiterator functions (GPU only):
    # bind the names to torch's own SiLU for the CPU run.
    import torch.nn.functional as _F
    ns.conv.ACT2FN = {"silu": _F.silu, "swish": _F.silu}
    ns.norm_gate = importlib.import_module("fla.modules.fused_norm_gate")
    return ns


# ---------------------------------------------------------------------------
# layer-level reference: infinitevl/infinitevl_standard/modeling_infinitevl.py
# ---------------------------------------------------------------------------
def load_reference_std(ns=None, carry_in_conv=True):
    """Import the reference's `infinitevl_standard.modeling_infinitevl` with the
    `fla` names it needs (std:52-54) bound to the vendored, interpreter-backed ops.

    Shims applied HERE (never to the reference): see SURVEY.md section 8c.
      * fla.layers.utils.{get_unpad_data,index_first_axis,pad_input}: dead code on
        this path (std:1223 nulls the mask) -> stubs that raise.
      * chunk/fused_recurrent: the call sites use fla>=0.4 time-major semantics;
        the vendored snapshot defaults to head_first=True -> wrappers passing
        head_first=False; for fp32 inputs the chunk wrapper calls
        chunk_gated_delta_rule_fwd directly (the public API asserts non-fp32,
        chunk.py:352).
      * ShortConvolution: built with use_fast_conv1d=False (causal_conv1d CUDA ext is
        absent).  With carry_in_conv=True a subclass emulates the pinned pip fla
        0.4.0 behaviour for (cache given, T>1): the cached inputs are carried in by
        convolving the vendored op over concat(cache[...,1:], x).
      * transformers 5.x: ROPE_INIT_FUNCTIONS['default'] re-added; CacheLayerMixin's
        new abstract get_max_length satisfied on the reference classes; an attention
        function 'ivl_band' registered that applies the sliding-window band to the
        reference's own eager_attention_forward.
    """
    import torch
    from einops import rearrange

    if ns is None:
        ns = load_reference_fla()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    # --- fla.layers.utils stubs
    layers = types.ModuleType("fla.layers")
    layers.__path__ = []
    lutils = types.ModuleType("fla.layers.utils")

    def _dead(*a, **k):
        raise RuntimeError("unpad path is dead code in InfiniteVL (std:1223)")

    lutils.get_unpad_data = lutils.index_first_axis = lutils.pad_input = _dead
    sys.modules["fla.layers"] = layers
    sys.modules["fla.layers.utils"] = lutils

    # --- fla.modules names
    fmods = sys.modules["fla.modules"]
    layernorm = importlib.import_module("fla.modules.layernorm")
    VendoredConv = ns.conv.ShortConvolution

    class ShortConvolutionRef(VendoredConv):
        def __init__(self, hidden_size, kernel_size, bias=False, activation="silu", **kw):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                super().__init__(hidden_size, kernel_size, bias=bias, activation=activation,
                                 use_fast_conv1d=False)

        def forward(self, x, mask=None, cache=None, output_final_state=False, cu_seqlens=None, **kw):
            B, T, D = x.shape
            W = self.kernel_size[0]
            if carry_in_conv and cache is not None and T > 1:
                ext = torch.cat([rearrange(cache[..., 1:], "b d w -> b w d").to(x.dtype), x], dim=1)
                y, _ = super().forward(ext, cache=None, output_final_state=False)
                full = torch.cat([rearrange(cache, "b d w -> b w d").to(x.dtype), x], dim=1)
                cache.copy_(rearrange(full[:, -W:], "b w d -> b d w"))
                return y[:, W - 1:], cache
            return super().forward(x, mask=mask, cache=cache, output_final_state=output_final_state,
                                   cu_seqlens=cu_seqlens, **kw)

    fmods.ShortConvolution = ShortConvolutionRef
    fmods.FusedRMSNormGated = ns.norm_gate.FusedRMSNormGated
    fmods.RMSNorm = layernorm.RMSNorm

    # --- fla.ops.gated_delta_rule wrappers (time-major)
    gdr = importlib.import_module("fla.ops.gated_delta_rule")

    def chunk_tm(q, k, v, g, beta, scale=None, initial_state=None, output_final_state=False,
                 cu_seqlens=None, use_qk_l2norm_in_kernel=False, **kw):
        assert cu_seqlens is None
        if q.dtype == torch.float32:
            if scale is None:
                scale = k.shape[-1] ** -0.5
            if use_qk_l2norm_in_kernel:
                q, k = ns.l2norm.l2norm_fwd(q.contiguous()), ns.l2norm.l2norm_fwd(k.contiguous())
            _, o, _, _, ht = ns.chunk.chunk_gated_delta_rule_fwd(
                q.contiguous(), k.contiguous(), v.contiguous(), g.contiguous(), beta.contiguous(), scale,
                initial_state, output_final_state, head_first=False)
            return o, ht
        return ns.chunk.chunk_gated_delta_rule(q, k, v, g, beta, scale=scale, initial_state=initial_state,
                                               output_final_state=output_final_state, head_first=False,
                                               use_qk_l2norm_in_kernel=use_qk_l2norm_in_kernel)

    def recurrent_tm(q, k, v, g, beta, scale=None, initial_state=None, output_final_state=False,
                     cu_seqlens=None, use_qk_l2norm_in_kernel=False, **kw):
        assert cu_seqlens is None
        return ns.recurrent.fused_recurrent_gated_delta_rule(
            q, k, v, g, beta, scale=scale, initial_state=initial_state,
            output_final_state=output_final_state, head_first=False,
            use_qk_l2norm_in_kernel=use_qk_l2norm_in_kernel)

    gdr.chunk_gated_delta_rule = chunk_tm
    gdr.fused_recurrent_gated_delta_rule = recurrent_tm
    ns.chunk_tm, ns.recurrent_tm = chunk_tm, recurrent_tm

    # --- transformers 5.x shims
    from transformers import modeling_rope_utils as mru

    if "default" not in mru.ROPE_INIT_FUNCTIONS:
        def _default_rope(config, device=None, **kw):
            base = config.rope_theta
            dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
            inv = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
            return inv, 1.0
        mru.ROPE_INIT_FUNCTIONS["default"] = _default_rope

    std = importlib.import_module("infinitevl.infinitevl_standard.modeling_infinitevl")
    for cls in (std.StaticSlidingWindowLayerPrealloc, std.StaticLinearLayerPrealloc):
        cls.get_max_length = lambda self: self.get_max_cache_shape()
        cls.__abstractmethods__ = frozenset()

    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS

    def ivl_band(module, query, key, value, attention_mask, dropout=0.0, scaling=None,
                 sliding_window=None, **kw):
        """Reference eager math (std:557-580) + FA2's bottom-right aligned causal band."""
        T, S = query.shape[2], key.shape[2]
        i = torch.arange(T)[:, None] + (S - T)
        j = torch.arange(S)[None, :]
        vis = j <= i
        if sliding_window is not None:
            vis = vis & (j > i - sliding_window)
        mask = torch.zeros(T, S, dtype=query.dtype).masked_fill(~vis, float("-inf"))[None, None]
        return std.eager_attention_forward(module, query, key, value, mask, scaling=scaling, dropout=0.0)

    ALL_ATTENTION_FUNCTIONS["ivl_band"] = ivl_band
    ns.std = std
    return ns
