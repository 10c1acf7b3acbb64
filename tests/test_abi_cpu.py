"""CPU (-m "not gpu"): the C-ABI library builds, loads and exports every symbol include/ivl_hip.h
declares; argument validation returns error codes without touching a GPU; the host-side mirror of the
reference interface (cache integer bookkeeping, error behaviour, parameter names) matches the
reference.  No compute call is made here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import cache as ocache



def _free_port():
    """A TCP port that is free right now on 127.0.0.1 (a pid-derived constant collided now and then with a socket of an
    earlier run still in TIME_WAIT)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]

@pytest.fixture(scope="session")
def lib():
    so = os.path.join(ROOT, "infinitevl_amd", "libivl_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    import infinitevl_amd
    return infinitevl_amd.load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ivl_hip.h")).read()
    declared = set(re.findall(r"\b(ivl_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("ivl_swa_args")
    assert len(declared) >= 13
    from infinitevl_amd._lib import EXPORTED_SYMBOLS
    assert declared == set(EXPORTED_SYMBOLS), declared ^ set(EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ivl_abi_version() == 11


def test_dynamic_symbol_table_is_exactly_the_header():
    """`nm -D` of the product library == the prototypes of include/ivl_hip.h: no debug hooks, no C++ symbols."""
    import subprocess
    from infinitevl_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    hdr = open(os.path.join(ROOT, "include", "ivl_hip.h")).read()
    declared = set(re.findall(r"^IVL_API [^\n(]*?\b(ivl_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert exported == declared, exported ^ declared
    src = "".join(open(os.path.join(ROOT, "infinitevl_amd", "csrc", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "infinitevl_amd", "csrc")) if f.endswith((".hip", ".h")))
    assert "getenv" not in src, "no environment knobs on the launch path"


def test_argument_validation_returns_codes(lib):
    from infinitevl_amd import _lib
    rc = lib.ivl_short_conv_fwd(None, None, None, None, None, 1, 1, 8, 4, 1, None)
    assert rc == _lib.IVL_ERR_INVALID_ARG and b"NULL" in lib.ivl_last_error()
    one = ctypes.c_void_p(0x1000)       # never dereferenced: validation fails first
    assert lib.ivl_short_conv_fwd(one, one, None, one, None, 1, 1, 8, 3, 1, None) == _lib.IVL_ERR_UNSUPPORTED
    assert lib.ivl_short_conv_fwd(one, one, None, one, None, 1, 1, 12, 4, 1, None) == _lib.IVL_ERR_UNSUPPORTED
    assert lib.ivl_short_conv_fwd(one, one, None, one, None, 0, 1, 8, 4, 1, None) == _lib.IVL_ERR_INVALID_ARG
    assert lib.ivl_gdn_recurrent_fwd(one, one, one, one, one, one, None, 2, None, 2, 1, 1, 2, 64, 256, 1.0, 1, None) \
        == _lib.IVL_ERR_UNSUPPORTED                                        # K != 128
    assert lib.ivl_gdn_chunk_fwd(one, one, one, one, one, one, None, 2, None, 2, 1, 100, 2, 128, 256, 1.0, 1, 0,
                                 one, 16, None) == _lib.IVL_ERR_WORKSPACE
    assert lib.ivl_gdn_chunk_fwd(one, one, one, one, one, one, None, 2, None, 2, 1, 100, 2, 128, 256, 1.0, 1, 1,
                                 one, 1 << 30, None) == _lib.IVL_ERR_INVALID_ARG                  # mma_dtype: bf16 or e4m3 only
    assert lib.ivl_rmsnorm_swish_gate_fwd(one, one, one, one, 4, 128, 1e-5, None) == _lib.IVL_ERR_UNSUPPORTED
    assert lib.ivl_mrope_fwd(one, one, one, one, 1, 1, 2, 1, 128, 16, 24, 23, None) == _lib.IVL_ERR_UNSUPPORTED
    assert lib.ivl_swa_fwd(None, None) == _lib.IVL_ERR_INVALID_ARG
    a = _lib.SwaArgs()
    a.q = a.k_new = a.v_new = a.o = 0x1000
    a.B, a.T, a.T_new, a.Hq, a.Hkv, a.d = 1, 4, 4, 16, 2, 64
    assert lib.ivl_swa_fwd(ctypes.byref(a), None) == _lib.IVL_ERR_UNSUPPORTED   # head_dim != 128
    a.d, a.Hq = 128, 3
    assert lib.ivl_swa_fwd(ctypes.byref(a), None) == _lib.IVL_ERR_INVALID_ARG   # Hq % Hkv
    assert lib.ivl_linear_small_m_fwd(one, one, None, one, 5, 16, 64, None) == _lib.IVL_ERR_UNSUPPORTED   # M > 4
    assert lib.ivl_linear_small_m_fwd(one, one, None, one, 1, 16, 60, None) == _lib.IVL_ERR_INVALID_ARG   # K % 8
    with pytest.raises(ValueError):
        _lib.check(_lib.IVL_ERR_INVALID_ARG)
    with pytest.raises(_lib.IvlError):
        _lib.check(_lib.IVL_ERR_LAUNCH)


def test_workspace_sizes(lib):
    per_chunk = 62464          # 54 MFMA fragment blocks of 1 KB + e^gamma / beta block + 6 blocks of Tu (u is made in the scan)
    assert lib.ivl_gdn_chunk_workspace_bytes(1, 256, 16, 128, 256) == 16 * 4 * per_chunk
    assert lib.ivl_gdn_chunk_workspace_bytes(2, 65, 3, 128, 256) == 2 * 3 * 2 * per_chunk
    # long calls are processed in 64-chunk segments: bounded workspace + fp32 state carry
    assert lib.ivl_gdn_chunk_workspace_bytes(1, 131072, 16, 128, 256) == 16 * 64 * per_chunk + 16 * 128 * 256 * 4
    assert lib.ivl_gdn_chunk_workspace_bytes(1, 256, 16, 64, 256) == 0
    assert lib.ivl_swa_workspace_bytes(1, 4096, 16, 128) <= (1 << 20) * 600
    assert lib.ivl_swa_workspace_bytes(1, 1, 16, 64) == 0
    # the 256-row attention path (ABI v11): rotated q + two linear copies [B, Hkv, C + T + 64, 128] bf16 (+ 256 B); 0 = the shape
    # does not qualify (T not a multiple of 256, fewer than 256 workgroups of 256 rows, a small ring, another head size)
    lin = 2 * (4095 + 4096 + 64) * 128 * 2
    assert lib.ivl_swa_ring256_workspace_bytes(1, 4096, 16, 2, 128, 4095) == 2 * lin + 4096 * 16 * 128 * 2 + 256
    assert lib.ivl_swa_ring256_workspace_bytes(8, 512, 16, 2, 128, 511) > 0
    for bad in ((1, 4000, 16, 2, 128, 4095), (1, 2048, 16, 2, 128, 4095), (1, 4096, 16, 2, 128, 300), (1, 4096, 16, 2, 64, 4095),
                (1, 4096, 16, 3, 128, 4095), (0, 4096, 16, 2, 128, 4095)):
        assert lib.ivl_swa_ring256_workspace_bytes(*bad) == 0, bad


def test_ops_refuse_cpu_tensors():
    from infinitevl_amd import ops
    z = torch.zeros(1, 4, 2, 128, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.chunk_gated_delta_rule(z, z, torch.zeros(1, 4, 2, 256, dtype=torch.bfloat16), torch.zeros(1, 4, 2),
                                   torch.zeros(1, 4, 2, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.swa_forward(z, z, z, window=8, scaling=1.0)


def test_product_does_not_import_oracle():
    """The product package must never route through the oracle (or any CPU fallback)."""
    code = "import sys; import infinitevl_amd, infinitevl_amd.harness, infinitevl_amd.dist; " \
           "print(int(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)))"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, check=True)
    assert out.stdout.strip().endswith("0")
    for fn in os.listdir(os.path.join(ROOT, "infinitevl_amd")):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "infinitevl_amd", fn)).read(), fn


# ---------------------------------------------------------------------------------------------
# cache: integer bookkeeping (bit-exact vs the reference traces) and error behaviour
# ---------------------------------------------------------------------------------------------
class _Cfg:
    num_key_value_heads, num_attention_heads, head_dim, hidden_size = 2, 4, 16, 64
    sliding_window, max_position_embeddings = 8, 4096
    num_linear_heads = num_linear_key_value_heads = 4
    linear_head_dim, conv_size, use_short_conv, expand_v = 16, 4, True, 2
    layer_types = ["sliding_attention", "linear_attention", "linear_attention", "linear_attention"]
    num_hidden_layers = 4


def test_swa_layer_counters_match_reference_traces():
    from infinitevl_amd.cache import StaticSlidingWindowLayerPrealloc
    z = load_golden("cache_traces")
    for key in sorted(k[:-6] for k in z if k.endswith("_steps")):
        steps, W = z[key + "_steps"].tolist(), int(z[key + "_W"])
        cfg = _Cfg()
        cfg.sliding_window = W
        layer = StaticSlidingWindowLayerPrealloc(config=cfg, batch_size=1, device="cpu", dtype=torch.bfloat16)
        assert layer.capacity == W - 1 and tuple(layer._buf_keys.shape) == (1, 2, W - 1, 16)
        ref = z[key + "_trace"].numpy()
        pos = 0
        for i, T in enumerate(steps):
            full_len = layer.size + T
            layer.advance(T)                       # host-side effect of one call (kernels not involved)
            kv_len, kv_off = layer.get_mask_sizes(torch.arange(pos, pos + T))
            assert (full_len, layer.size, layer.cumulative_length, kv_len, kv_off) == tuple(ref[i].tolist()), (key, i)
            assert layer.get_seq_length() == layer.cumulative_length
            pos += T
    layer = StaticSlidingWindowLayerPrealloc(config=_Cfg(), batch_size=2, device="cpu", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        layer.batch_repeat_interleave(2)
    with pytest.raises(RuntimeError):
        layer.batch_select_indices(torch.arange(3))
    layer.batch_repeat_interleave(1)
    layer.advance(9)
    with pytest.raises(ValueError, match="Cropping is forbidden"):
        layer.crop(3)
    bad = _Cfg()
    bad.sliding_window, bad.max_position_embeddings = 0, 0
    with pytest.raises(ValueError):
        StaticSlidingWindowLayerPrealloc(config=bad, batch_size=1)


def test_swa_ring_chronological_view_matches_reference_tail():
    """`keys` must present the cached tokens oldest-first exactly like the reference's linear tail
    buffer (fixture: positions held after each call of the probed trace)."""
    from infinitevl_amd.cache import StaticSlidingWindowLayerPrealloc
    z = load_golden("cache_traces")
    layer = StaticSlidingWindowLayerPrealloc(config=_Cfg(), batch_size=1, device="cpu", dtype=torch.float32)
    pos, off = 0, 0
    for T, n in zip([5, 1, 1, 1, 6, 1, 20], z["W8_tail_lengths"].tolist()):
        for t in range(max(0, T - layer.capacity), T):          # what ivl_swa_cache_append writes
            layer._buf_keys[:, :, (pos + t) % layer.capacity, :] = float(pos + t)
        layer.advance(T)
        pos += T
        got = layer.keys[0, 0, :, 0].numpy()
        assert np.array_equal(got, z["W8_tail_positions"].numpy()[off:off + n])
        off += n


def test_swa_crop_matches_reference_trace():
    """std:192-213 through the reference-generated trace: crop keeps the LAST new_size tokens and restarts the counters."""
    from infinitevl_amd.cache import StaticSlidingWindowLayerPrealloc
    z = load_golden("cache_traces")
    layer = StaticSlidingWindowLayerPrealloc(config=_Cfg(), batch_size=1, device="cpu", dtype=torch.float32)
    pos, off = 0, 0
    for op, n, ctr in zip(z["crop_ops"].tolist(), z["crop_tail_lengths"].tolist(), z["crop_counters"].tolist()):
        if op < 500:
            for t in range(op):                      # what ivl_swa_cache_append writes (window not full: slot == position)
                layer._buf_keys[:, :, (layer.cumulative_length + t) % layer.capacity, :] = float(pos + t)
            layer.advance(op)
            layer._pos_dev += op
            pos += op
        else:
            layer.crop(op - 1000)
        assert [layer.size, layer.cumulative_length] == ctr and int(layer._pos_dev) == ctr[1]
        assert np.array_equal(layer.keys[0, 0, :, 0].numpy(), z["crop_tail_positions"].numpy()[off:off + n]), op
        off += n


def test_linear_layer_protocol_and_errors():
    from infinitevl_amd.cache import StaticLinearLayerPrealloc
    layer = StaticLinearLayerPrealloc(config=_Cfg(), batch_size=2, device="cpu", dtype=torch.float32, zero_init=True)
    assert tuple(layer.recurrent_state.shape) == (2, 4, 16, 32)
    assert tuple(layer.conv_state_v.shape) == (2, 4 * 32, 4)
    ref = ocache.LinearCounters()
    (cq, ck, cv), rec = layer.update(cache_kwargs={"op": "get"})
    assert cq is None and rec is None and ref.get() is False           # first call returns Nones (std:298-300)
    new = torch.ones(2, 4, 16, 32)
    layer.update(conv_state=(torch.ones(2, 64, 4), None, None), recurrent_state=new, cache_kwargs={"op": "set", "delta_len": 7})
    ref.set(7)
    assert layer.seq_len == ref.seq_len == 7 and torch.equal(layer.recurrent_state, new)
    (cq, _, _), rec = layer.update(cache_kwargs={"op": "get"})
    assert rec is layer.recurrent_state and float(cq.sum()) == 2 * 64 * 4
    with pytest.raises(RuntimeError, match="recurrent_state shape changed"):
        layer.update(recurrent_state=torch.ones(2, 4, 16, 16), cache_kwargs={"op": "set"})
    with pytest.raises(RuntimeError, match="conv_q shape changed"):
        layer.update(conv_state=(torch.ones(2, 60, 4), None, None), cache_kwargs={"op": "set"})
    assert layer.get_mask_sizes(torch.arange(3)) == (10, 0)
    layer.crop(-2)
    assert layer.seq_len == 5


def test_aggregate_cache_dispatch_clone_copy():
    from infinitevl_amd.cache import (StaticCachePrealloc, StaticLinearLayerPrealloc,
                                      StaticSlidingWindowLayerPrealloc)
    c = StaticCachePrealloc(config=_Cfg(), batch_size=1, device="cpu", dtype=torch.bfloat16, zero_init=True)
    assert [type(l) for l in c.layers] == [StaticSlidingWindowLayerPrealloc] + [StaticLinearLayerPrealloc] * 3
    for name in ("is_sliding", "_buf_keys", "_buf_values", "keys", "values", "size", "cumulative_length", "capacity"):
        assert hasattr(c.layers[0], name)          # attribute names read by the demo's clone (demo:123-146)
    for name in ("conv_state_q", "conv_state_k", "conv_state_v", "recurrent_state", "seq_len", "start"):
        assert hasattr(c.layers[1], name)          # demo:147-158
    c.advance(5)
    c.layers[1].recurrent_state.fill_(3.0)
    d = c.clone()
    d.advance(2)
    d.layers[1].recurrent_state.fill_(4.0)
    assert c.get_seq_length() == 5 and d.get_seq_length() == 7
    assert float(c.layers[1].recurrent_state[0, 0, 0, 0]) == 3.0
    c.copy_from(d)
    assert c.get_seq_length() == 7 and float(c.layers[1].recurrent_state[0, 0, 0, 0]) == 4.0
    assert c.layers[1].recurrent_state.data_ptr() != d.layers[1].recurrent_state.data_ptr()


def test_module_parameter_names_match_reference_checkpoint_layout():
    """state_dict keys/shapes of the drop-in modules == the reference modules' (fixture holds the
    reference state_dict of a tiny config: std:1019-1022, 1161-1213)."""
    from infinitevl_amd.harness import InfiniteVLDecoderLayer, InfiniteVLTextConfig
    z = load_golden("tiny_stack")
    lt = [str(x) for x in z["layer_types"]]
    cfg = InfiniteVLTextConfig(vocab_size=97, hidden_size=64, intermediate_size=96, num_hidden_layers=4,
                               num_attention_heads=4, num_key_value_heads=2, head_dim=16, sliding_window=8,
                               layer_types=lt, num_linear_heads=4, num_linear_key_value_heads=4, linear_head_dim=16,
                               rope_scaling={"mrope_section": [2, 3, 3]})
    for i in range(4):
        ours = {k: tuple(v.shape) for k, v in InfiniteVLDecoderLayer(cfg, i).state_dict().items()}
        ref = {k[len(f"w.layers.{i}."):]: tuple(v.shape) for k, v in z.items() if k.startswith(f"w.layers.{i}.")}
        assert ours == ref, (i, set(ours) ^ set(ref))


def test_dynamic_layer_fallback_and_total_dispatch():
    """strm:67-157, 548-550: a layer type that is neither sliding nor linear gets a growing K/V cache, so
    `cache.layers[i]` exists for every decoder layer."""
    from infinitevl_amd.cache import DynamicLayer, StaticCachePrealloc, StaticLinearLayerPrealloc, \
        StaticSlidingWindowLayerPrealloc
    from infinitevl_amd.harness import InfiniteVLTextConfig
    cfg = InfiniteVLTextConfig(vocab_size=97, hidden_size=64, intermediate_size=96, num_hidden_layers=3,
                               num_attention_heads=4, num_key_value_heads=2, head_dim=16, sliding_window=8,
                               layer_types=["sliding_attention", "full_attention", "linear_attention"],
                               num_linear_heads=4, num_linear_key_value_heads=4, linear_head_dim=16)
    cache = StaticCachePrealloc(config=cfg, batch_size=1, device="cpu", dtype=torch.float32)
    assert [type(layer) for layer in cache.layers] == [StaticSlidingWindowLayerPrealloc, DynamicLayer,
                                                        StaticLinearLayerPrealloc]
    dyn = cache.layers[1]
    assert dyn.get_seq_length() == 0 and dyn.get_max_cache_shape() == -1
    k1, v1 = torch.randn(1, 2, 5, 16), torch.randn(1, 2, 5, 16)
    k2, v2 = torch.randn(1, 2, 3, 16), torch.randn(1, 2, 3, 16)
    fk, fv = cache.update(1, k1, v1)
    assert torch.equal(fk, k1) and dyn.get_seq_length() == 5
    fk, fv = cache.update(1, k2, v2)
    assert torch.equal(fk, torch.cat([k1, k2], -2)) and torch.equal(fv, torch.cat([v1, v2], -2))
    assert dyn.get_mask_sizes(torch.arange(3)) == (8, 0)
    twin = cache.clone().layers[1]
    dyn.crop(6)
    assert dyn.get_seq_length() == 6 and twin.get_seq_length() == 8 and torch.equal(dyn.keys, fk[..., :6, :])
    dyn.crop(-2)
    assert dyn.get_seq_length() == 4
    dyn.batch_repeat_interleave(3)
    assert dyn.keys.shape[0] == 3
    dyn.batch_select_indices(torch.tensor([0]))
    assert dyn.keys.shape[0] == 1
    cache.reset()
    assert dyn.get_seq_length() == 0


def _tiny_cfg(z):
    from infinitevl_amd.harness import InfiniteVLTextConfig
    lt = [str(x) for x in z["layer_types"]]
    return InfiniteVLTextConfig(vocab_size=97, hidden_size=64, intermediate_size=96, num_hidden_layers=4,
                                num_attention_heads=4, num_key_value_heads=2, head_dim=16, sliding_window=8,
                                layer_types=lt, num_linear_heads=4, num_linear_key_value_heads=4, linear_head_dim=16,
                                rope_scaling={"mrope_section": [2, 3, 3]})


@pytest.mark.parametrize("prefix", ["model.language_model.", "model."])
def test_load_reference_checkpoint_layouts(tmp_path, prefix):
    """The reference checkpoint's tensor names (current `model.language_model.*` and legacy `model.*` layouts, vision
    tower and tied lm_head alongside) load into the drop-in text stack; values = the reference state_dict fixture."""
    from safetensors.torch import save_file
    from infinitevl_amd.harness import InfiniteVLTextStack, load_reference_checkpoint
    z = load_golden("tiny_stack")
    ref = {k[2:]: v.clone() for k, v in z.items() if k.startswith("w.")}
    ckpt = {prefix + k: v.contiguous() for k, v in ref.items()}
    vis = ("model.visual." if prefix != "model." else "visual.") + "blocks.0.attn.qkv.weight"
    ckpt[vis] = torch.zeros(3, 3)
    ckpt["lm_head.weight"] = ref["embed_tokens.weight"].clone()
    half = len(ckpt) // 2
    names = sorted(ckpt)
    save_file({k: ckpt[k] for k in names[:half]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: ckpt[k] for k in names[half:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    stack = InfiniteVLTextStack(_tiny_cfg(z))
    skipped = load_reference_checkpoint(stack, str(tmp_path))
    assert sorted(skipped) == sorted([vis, "lm_head.weight"])
    sd = stack.state_dict()
    for k, v in ref.items():
        assert torch.equal(sd[k].float(), v.float()), k
    # a checkpoint that lacks a text tensor, or carries a wrong shape, is refused
    bad = dict(ckpt)
    del bad[prefix + "norm.weight"]
    save_file(bad, str(tmp_path / "bad.safetensors"))
    with pytest.raises(KeyError):
        load_reference_checkpoint(InfiniteVLTextStack(_tiny_cfg(z)), str(tmp_path / "bad.safetensors"))
    bad = dict(ckpt)
    bad[prefix + "norm.weight"] = torch.zeros(3)
    save_file(bad, str(tmp_path / "bad2.safetensors"))
    with pytest.raises(ValueError):
        load_reference_checkpoint(InfiniteVLTextStack(_tiny_cfg(z)), str(tmp_path / "bad2.safetensors"))


def test_text_config_from_reference_config_json():
    """The shipped InfiniteVL-3B config.json values (top-level text fields + ignored vision / token-id keys)."""
    from infinitevl_amd.harness import InfiniteVLTextConfig
    cfg = {"architectures": ["InfiniteVLQwen2_5_VLForConditionalGeneration"], "hidden_size": 2048, "intermediate_size": 11008,
           "num_hidden_layers": 36, "num_attention_heads": 16, "num_key_value_heads": 2, "rms_norm_eps": 1e-06,
           "rope_theta": 1000000.0, "sliding_window": 8192, "tie_word_embeddings": True, "vocab_size": 151936,
           "rope_scaling": {"mrope_section": [16, 24, 24], "rope_type": "default", "type": "default"},
           "vision_config": {"hidden_size": 1280}, "image_token_id": 151655, "text_config": {"sliding_window": 4096}}
    c = InfiniteVLTextConfig.from_hf_config(cfg)
    assert (c.hidden_size, c.num_hidden_layers, c.num_key_value_heads, c.vocab_size) == (2048, 36, 2, 151936)
    assert c.sliding_window == 4096                      # text_config overrides the top level
    assert c.layer_types[:5] == ["sliding_attention", "linear_attention", "linear_attention", "linear_attention",
                                 "sliding_attention"] and len(c.layer_types) == 36
    assert c == InfiniteVLTextConfig(sliding_window=4096, rope_scaling=cfg["rope_scaling"])


# ---------------------------------------------------------------------------------------------
# multi-GPU host logic on CPU: gloo, world_size 2
# ---------------------------------------------------------------------------------------------
def test_shard_batch_properties():
    from infinitevl_amd.dist import shard_batch
    for world in (1, 2, 3, 8):
        for gb in (0, 1, 7, 8, 19):
            spans = [shard_batch(gb, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == gb
            assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_batch(4, 2, 2)


def _gloo_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from infinitevl_amd import dist as ivd
    r, w, _ = ivd.init_distributed("gloo")
    counts = [ivd.shard_batch(5, i, w)[1] for i in range(w)]          # ragged: 3 + 2
    first, cnt = ivd.shard_batch(5, r, w)
    local = torch.arange(first, first + cnt, dtype=torch.float32)[:, None] * torch.ones(1, 11)
    full = ivd.gather_last_logits(local, counts)
    ivd.barrier()
    mx = ivd.max_over_ranks(1.0 + r, torch.device("cpu"))
    info = ivd.describe_ranks(torch.device("cpu"))
    assert info["backend"] == "gloo" and info["world"] == w and [d["rank"] for d in info["devices"]] == list(range(w))
    q.put((r, full[:, 0].tolist(), mx))
    torch.distributed.destroy_process_group()


def test_gloo_world2_gather_and_max():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    for r, col, mx in res:
        assert col == [0.0, 1.0, 2.0, 3.0, 4.0] and mx == 2.0


# ---- sequence-parallel prefill: schedule + per-layer state hand-off over gloo -------------------
class _ToyLayerCache:
    """A layer state with the cache contract dist.sequence_parallel_prefill relies on."""

    def __init__(self, B, H):
        self.state = torch.zeros(B, H, dtype=torch.float64)
        self.ring = torch.zeros(B, 3, H, dtype=torch.float64)        # last 3 inputs (a sliding window)
        self.seen = 0

    def carried_tensors(self):
        return [self.state, self.ring]

    def import_carried(self, seen_tokens):
        self.seen = int(seen_tokens)


class _ToyCache:
    def __init__(self, L, B, H):
        self.layers = [_ToyLayerCache(B, H) for _ in range(L)]


class _ToyStack:
    """Token-recurrent toy layers whose output depends on ALL earlier tokens (decayed state) and on the last 3
    inputs (window) and the absolute position: any lost / misordered / stale hand-off changes the result."""

    def __init__(self, L):
        self.decay = [0.5 + 0.1 * i for i in range(L)]

    def __call__(self, inputs_embeds, position_ids, past_key_values, logits_to_keep=0, layer_hooks=None):
        x = inputs_embeds.clone()
        for i, a in enumerate(self.decay):
            if layer_hooks is not None:
                layer_hooks[0](i)
            c = past_key_values.layers[i]
            out = torch.empty_like(x)
            for t in range(x.shape[1]):
                c.state = a * c.state + x[:, t]
                win = c.ring.sum(1)
                out[:, t] = x[:, t] + c.state + 0.25 * win + 1e-3 * position_ids[0, :, t, None].double()
                c.ring = torch.cat([c.ring[:, 1:], x[:, t:t + 1]], dim=1)
            # the contract is in-place state: write back into the tensors carried_tensors() returned at creation
            if layer_hooks is not None:
                layer_hooks[1](i)
            x = out
        return x, None


def _sp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from infinitevl_amd import dist as ivd
    ivd.init_distributed("gloo")
    B, H, L, total = 2, 5, 4, 150
    torch.manual_seed(0)
    xs = torch.randn(B, total, H, dtype=torch.float64)
    first, last = ivd.segment_bounds(total, rank, world, multiple=64)          # 128 + 22
    cache = _ToyCache(L, B, H)
    h, _ = ivd.sequence_parallel_prefill(_ToyStack(L), xs[:, first:last], cache, first)
    # numpy, not tensors: a tensor travels through the queue as a shared-memory file descriptor that the RECEIVER fetches
    # from this process - which may have exited by then (EOFError in 1 of ~12 runs)
    q.put((rank, first, last, h.numpy(), [c.state.numpy().copy() for c in cache.layers], [c.seen for c in cache.layers]))
    torch.distributed.destroy_process_group()


def test_gloo_world2_sequence_parallel_prefill_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    # single-process reference: the whole sequence through the same stack
    B, H, L, total = 2, 5, 4, 150
    torch.manual_seed(0)
    xs = torch.randn(B, total, H, dtype=torch.float64)
    cache = _ToyCache(L, B, H)
    pos = torch.arange(total)[None, None].expand(3, B, total)
    ref, _ = _ToyStack(L)(xs, pos, cache)
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 128, 128, 150)
    got = torch.cat([torch.from_numpy(res[0][3]), torch.from_numpy(res[1][3])], dim=1)
    assert torch.equal(got, ref)
    for s_ref, s_got in zip([c.state for c in cache.layers], res[1][4]):       # last rank ends with the full state
        assert torch.equal(s_ref, torch.from_numpy(s_got))
    assert res[1][5] == [128] * L and res[0][5] == [0] * L


def test_segment_bounds_properties():
    from infinitevl_amd.dist import segment_bounds
    for world in (1, 2, 3, 8):
        for total in (0, 1, 64, 65, 1000, 131072):
            spans = [segment_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert all(a % 64 == 0 or a == total for a, _ in spans)           # boundaries are GDN chunk boundaries
    with pytest.raises(ValueError):
        segment_bounds(10, 2, 2)
