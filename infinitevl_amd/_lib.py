"""ctypes binding of libivl_hip.so (the C ABI declared in include/ivl_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C infinitevl_amd/csrc`.
There is NO fallback: if the shared object is missing, importing an operator raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libivl_hip.so")

IVL_BF16, IVL_F32, IVL_FP8_E4M3 = 0, 2, 3
IVL_OK = 0
IVL_GDN_SYNC_BYTES = 16384
IVL_GDN_RESIDENT_QUERY = -2147483648      # ivl_gdn_resident_blocks: a pure read
IVL_ERR_INVALID_ARG, IVL_ERR_UNSUPPORTED, IVL_ERR_WORKSPACE, IVL_ERR_LAUNCH, IVL_ERR_SYNC = -1, -2, -3, -4, -5

EXPORTED_SYMBOLS = (
    "ivl_abi_version", "ivl_last_error",
    "ivl_gdn_recurrent_fwd", "ivl_gdn_chunk_workspace_bytes", "ivl_gdn_chunk_fwd", "ivl_gdn_gate_fwd",
    "ivl_short_conv_fwd", "ivl_rmsnorm_swish_gate_fwd", "ivl_mrope_fwd",
    "ivl_swa_workspace_bytes", "ivl_swa_fwd", "ivl_swa_cache_append", "ivl_counter_add",
    "ivl_gdn_prologue_fwd", "ivl_rmsnorm_swish_gate_strided_fwd", "ivl_mrope_strided_fwd",
    "ivl_add_rmsnorm_fwd", "ivl_silu_mul_fwd", "ivl_linear_small_m_fwd",
    "ivl_linear_swiglu_small_m_fwd", "ivl_gdn_decode_step_fwd", "ivl_gdn_chunk_fused_fwd", "ivl_rope_tables_fwd",
    "ivl_vision_attn_workspace_bytes", "ivl_vision_attn_fwd", "ivl_norm_linear_small_m_fwd",
    "ivl_gdn_sync_status", "ivl_gdn_sync_reset", "ivl_gdn_resident_blocks",
    "ivl_short_conv_bias_fwd", "ivl_rmsnorm_swish_gate_res_fwd", "ivl_gdn_recurrent_f16_fwd",
    "ivl_gdn_decode_split_fwd", "ivl_gdn_out_linear_small_m_fwd", "ivl_swa_ring256_workspace_bytes",
)


class SwaArgs(Structure):
    """struct ivl_swa_args (include/ivl_hip.h)."""
    _fields_ = [
        ("q", c_void_p), ("k_new", c_void_p), ("v_new", c_void_p), ("k_cache", c_void_p), ("v_cache", c_void_p),
        ("o", c_void_p),
        ("q_sb", c_int64), ("q_st", c_int64), ("q_sh", c_int64),
        ("kn_sb", c_int64), ("kn_st", c_int64), ("kn_sh", c_int64),
        ("B", c_int), ("T", c_int), ("T_new", c_int), ("Hq", c_int), ("Hkv", c_int), ("d", c_int),
        ("cache_capacity", c_int), ("window", c_int),
        ("pos", c_int64), ("pos_dev", c_void_p),
        ("scaling", c_float),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("rope_cos", c_void_p), ("rope_sin", c_void_p), ("rope_s0", c_int), ("rope_s1", c_int),
        ("mma_dtype", c_int), ("append_new", c_int),
        ("pos_min", c_int64),
    ]


class IvlError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libivl_hip error {code}: {msg}")
        self.code = code


_lib = None


def load(path: str = None) -> ctypes.CDLL:
    """Load the shared object once and declare every prototype.  `path` (developer tools only: the
    instrumented build libivl_hip_trace.so) must be given before the first operator call."""
    global _lib
    if _lib is not None:
        return _lib
    if path is not None:
        globals()["LIB_PATH"] = path
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found. The MI355X kernels are mandatory (there is no CPU or PyTorch fallback): "
            "build them with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C infinitevl_amd/csrc`.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i, f, sz, i64 = c_void_p, c_int, c_float, c_size_t, c_int64
    lib.ivl_abi_version.restype = i
    lib.ivl_abi_version.argtypes = []
    lib.ivl_last_error.restype = c_char_p
    lib.ivl_last_error.argtypes = []
    lib.ivl_gdn_recurrent_fwd.restype = i
    lib.ivl_gdn_recurrent_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, vp, i, i, i, i, i, i, f, i, vp]
    lib.ivl_gdn_chunk_workspace_bytes.restype = sz
    lib.ivl_gdn_chunk_workspace_bytes.argtypes = [i, i, i, i, i]
    lib.ivl_gdn_chunk_fwd.restype = i
    lib.ivl_gdn_chunk_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, vp, i, i, i, i, i, i, f, i, i, vp, sz, vp]
    lib.ivl_gdn_chunk_fused_fwd.restype = i
    lib.ivl_gdn_chunk_fused_fwd.argtypes = ([vp, c_int64, i, i, i, i, i] + [vp] * 9 + [vp, vp, vp, vp, i, vp, i] +
                                            [i, i, i, i, i, i, f, i, vp, sz, vp, vp])
    if path is None or hasattr(lib, "ivl_gdn_sync_status"):     # (a developer A/B against an older build lacks the v8 entries)
        lib.ivl_gdn_sync_status.restype = i
        lib.ivl_gdn_sync_status.argtypes = [vp, vp]
        lib.ivl_gdn_sync_reset.restype = i
        lib.ivl_gdn_sync_reset.argtypes = [vp, vp]
        lib.ivl_gdn_resident_blocks.restype = i
        lib.ivl_gdn_resident_blocks.argtypes = [i]
        lib.ivl_short_conv_bias_fwd.restype = i
        lib.ivl_short_conv_bias_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        lib.ivl_rmsnorm_swish_gate_res_fwd.restype = i
        lib.ivl_rmsnorm_swish_gate_res_fwd.argtypes = [vp, vp, vp, vp, i, vp, i, vp, i, i, f, vp]
        lib.ivl_gdn_recurrent_f16_fwd.restype = i
        lib.ivl_gdn_recurrent_f16_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, vp, i, i, i, i, i, i, f, i, vp]
    lib.ivl_rope_tables_fwd.restype = i
    lib.ivl_rope_tables_fwd.argtypes = [vp, vp, vp, vp, i, i, f, vp]
    lib.ivl_vision_attn_workspace_bytes.restype = sz
    lib.ivl_vision_attn_workspace_bytes.argtypes = [i, i, i, i]
    lib.ivl_vision_attn_fwd.restype = i
    lib.ivl_vision_attn_fwd.argtypes = [vp, vp, vp, vp] + [c_int64] * 8 + [vp, i, i, i, i, i, f, vp, vp, vp, sz, vp]
    lib.ivl_gdn_gate_fwd.restype = i
    lib.ivl_gdn_gate_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, vp]
    lib.ivl_short_conv_fwd.restype = i
    lib.ivl_short_conv_fwd.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, vp]
    lib.ivl_rmsnorm_swish_gate_fwd.restype = i
    lib.ivl_rmsnorm_swish_gate_fwd.argtypes = [vp, vp, vp, vp, i, i, f, vp]
    lib.ivl_mrope_fwd.restype = i
    lib.ivl_mrope_fwd.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, i, i, vp]
    lib.ivl_swa_workspace_bytes.restype = sz
    lib.ivl_swa_workspace_bytes.argtypes = [i, i, i, i]
    lib.ivl_swa_ring256_workspace_bytes.restype = sz
    lib.ivl_swa_ring256_workspace_bytes.argtypes = [i, i, i, i, i, i]
    lib.ivl_swa_fwd.restype = i
    lib.ivl_swa_fwd.argtypes = [POINTER(SwaArgs), vp]
    lib.ivl_swa_cache_append.restype = i
    lib.ivl_swa_cache_append.argtypes = [vp, vp, i64, i64, i64, vp, vp, i, i, i, i, i, i64, vp, vp, vp, i, i, vp]
    lib.ivl_counter_add.restype = i
    lib.ivl_counter_add.argtypes = [vp, i64, vp]
    lib.ivl_gdn_prologue_fwd.restype = i
    lib.ivl_gdn_prologue_fwd.argtypes = [vp, i64, i, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                         vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, vp]
    lib.ivl_rmsnorm_swish_gate_strided_fwd.restype = i
    lib.ivl_rmsnorm_swish_gate_strided_fwd.argtypes = [vp, vp, i64, i, vp, vp, i, i, f, vp]
    lib.ivl_mrope_strided_fwd.restype = i
    lib.ivl_mrope_strided_fwd.argtypes = [vp, vp, i64, i64, vp, vp, i, i, i, i, i, i, i, i, vp]
    lib.ivl_add_rmsnorm_fwd.restype = i
    lib.ivl_add_rmsnorm_fwd.argtypes = [vp, vp, vp, vp, vp, i, i, f, vp]
    lib.ivl_silu_mul_fwd.restype = i
    lib.ivl_silu_mul_fwd.argtypes = [vp, vp, i64, i, vp]
    lib.ivl_linear_small_m_fwd.restype = i
    lib.ivl_linear_small_m_fwd.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    lib.ivl_gdn_decode_step_fwd.restype = i
    lib.ivl_gdn_decode_step_fwd.argtypes = [vp, i64, i, i, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, f, vp, i,
                                            vp, i, i, i, i, f, vp]
    if path is None or hasattr(lib, "ivl_gdn_decode_split_fwd"):  # (a developer A/B against a pre-v10 build lacks them)
        lib.ivl_gdn_decode_split_fwd.restype = i
        lib.ivl_gdn_decode_split_fwd.argtypes = [vp, i64, i, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, vp, i, i, i, i, f, vp]
        lib.ivl_gdn_out_linear_small_m_fwd.restype = i
        lib.ivl_gdn_out_linear_small_m_fwd.argtypes = [vp, vp, i64, vp, f, i, vp, i64, i, i, vp, vp, vp, vp, vp, i, i, i, vp]
    lib.ivl_linear_swiglu_small_m_fwd.restype = i
    lib.ivl_linear_swiglu_small_m_fwd.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    lib.ivl_norm_linear_small_m_fwd.restype = i
    lib.ivl_norm_linear_small_m_fwd.argtypes = [vp, vp, vp, f, vp, vp, vp, vp, i, i, i, i, vp]
    _lib = lib
    # the one environment switch, on the Python side: IVL_GDN_RESIDENT_BLOCKS=0 forces the two-launch form of the fused GDN call
    env = os.environ.get("IVL_GDN_RESIDENT_BLOCKS", "")
    if env.strip() and hasattr(lib, "ivl_gdn_resident_blocks"):
        lib.ivl_gdn_resident_blocks(int(env))
    return lib


def check(rc: int) -> None:
    """Map a C status to the Python exception types the reference raises
    (SURVEY.md section 8b "Error conventions")."""
    if rc == IVL_OK:
        return
    msg = load().ivl_last_error().decode("utf-8", "replace")
    if rc in (IVL_ERR_INVALID_ARG, IVL_ERR_UNSUPPORTED):
        raise ValueError(f"libivl_hip ({rc}): {msg}")
    raise IvlError(rc, msg)
