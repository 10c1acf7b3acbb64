"""Operator-level host API: the same names, argument meaning and error behaviour as the
third-party operators the reference calls (SURVEY.md section 8b "Operator-level contract"), backed by
the gfx950 kernels of libivl_hip.so.  PyTorch is used for device memory and streams only.

    chunk_gated_delta_rule / fused_recurrent_gated_delta_rule   <- fla.ops.gated_delta_rule  (std:1297-1320)
    ShortConvolution, FusedRMSNormGated, RMSNorm                <- fla.modules               (std:53, 1187-1213)
    get_unpad_data, index_first_axis, pad_input                 <- fla.layers.utils          (std:52, 1253-1258, 1344-1345)
    swa_attention_interface                                     <- ALL_ATTENTION_FUNCTIONS["flash_attention_2"] (std:1097-1108)
    gdn_gate, apply_mrope_inplace                               <- torch glue at std:1293-1294 / std:1057-1064

`std:` = infinitevl/infinitevl_standard/modeling_infinitevl.py of the reference.
There is no CPU path: tensors must live on a ROCm device and the shared library must be built.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
import weakref
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import IVL_BF16, IVL_F32, IVL_FP8_E4M3, SwaArgs

_DT_CODE = {torch.bfloat16: IVL_BF16, torch.float32: IVL_F32}
_MMA_CODE = {None: IVL_BF16, "bf16": IVL_BF16, torch.bfloat16: IVL_BF16, "fp8": IVL_FP8_E4M3, "fp8_e4m3": IVL_FP8_E4M3,
             getattr(torch, "float8_e4m3fn", "fp8_e4m3"): IVL_FP8_E4M3}


def mma_code(mma_dtype) -> int:
    """Operand format of the MFMA products: None / "bf16" (the reference's precision) or "fp8_e4m3"
    (BASELINE.json configs[4]: e4m3 operands, fp32 accumulation and state)."""
    try:
        return _MMA_CODE[mma_dtype]
    except KeyError:
        raise ValueError(f"mma_dtype must be None, 'bf16' or 'fp8_e4m3' (got {mma_dtype!r})") from None


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "infinitevl_amd ops run only on an MI355X (ROCm) device; got a CPU tensor. "
                "There is deliberately no CPU fallback in the product path.")


# ---------------------------------------------------------------------------------------------
# workspace: one growable scratch buffer per (device, tag)
# ---------------------------------------------------------------------------------------------
_WORKSPACES: Dict[Tuple[int, str], torch.Tensor] = {}
_GRAPH_PINNED: Dict[Tuple[int, str], bool] = {}
_RETIRED: list = []          # buffers a captured hipGraph may still address: never returned to the allocator


def get_workspace(nbytes: int, device: torch.device, tag: str = "main") -> torch.Tensor:
    """Caller-owned scratch handed to the C ABI (the library never allocates).  One buffer per (device, tag), used by
    one stream at a time (the package issues everything on the current stream; a hipGraph capture re-uses the buffer of
    its warm-up, ordered by the capture's own stream joins -- callers that overlap several streams pass distinct tags).
    It only ever grows -- and a buffer that was handed out while a stream was being captured is baked into a hipGraph,
    so when a later eager call outgrows it the old buffer is RETIRED (kept alive for the life of the process), never
    freed: replaying the graph stays valid.  Growth during capture itself is refused -- run one warm-up step first."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, tag)
    capturing = torch.cuda.is_current_stream_capturing()
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        if capturing:
            raise RuntimeError("workspace would have to grow during hipGraph capture; run a warm-up step first")
        if ws is not None and _GRAPH_PINNED.get(key):
            _RETIRED.append(ws)
            _GRAPH_PINNED[key] = False
        # geometric growth bounds the number of retired (graph-pinned) buffers of a run whose calls keep growing; above 256 MiB
        # it is +25 % (a retired multi-GB buffer stays alive beside its successor: doubling there can exhaust the device)
        grown = 0
        if ws is not None:
            grown = 2 * ws.numel() if ws.numel() < (256 << 20) else ws.numel() + ws.numel() // 4
        ws = torch.empty(max(int(nbytes), grown, 1 << 20), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = ws
    if capturing:
        _GRAPH_PINNED[key] = True
    return ws


# ---------------------------------------------------------------------------------------------
# Gated DeltaNet
# ---------------------------------------------------------------------------------------------
def _gdn_common(q, k, v, g, beta, scale, initial_state, output_final_state, cu_seqlens, final_state_out):
    _need_gpu(q, k, v, g, beta, initial_state)
    assert q.dtype == k.dtype == v.dtype, "q, k, v must share a dtype"
    assert len(beta.shape) == 3, "beta must be of shape [B, T, H]"                       # chunk.py:353
    if q.dtype not in (torch.bfloat16, torch.float16):                                    # chunk.py:352 refuses fp32 only
        raise ValueError(f"infinitevl_amd GDN kernels take bf16 (or fp16) activations, got {q.dtype}")
    B, T, H, K = k.shape
    V = v.shape[-1]
    NS = B                                                                                # number of sequences = number of states
    if cu_seqlens is not None:
        if q.shape[0] != 1:                                                               # chunk.py:355-360
            raise ValueError(
                f"The batch size is expected to be 1 rather than {q.shape[0]} when using `cu_seqlens`."
                f"Please flatten variable-length inputs before processing.")
        NS = len(cu_seqlens) - 1
        if initial_state is not None and initial_state.shape[0] != NS:                    # chunk.py:365-369
            raise ValueError(
                f"The number of initial states is expected to be equal to the number of input sequences, "
                f"i.e., {NS} rather than {initial_state.shape[0]}.")
    if not (q.shape[2] == k.shape[2] == v.shape[2] == g.shape[2] == beta.shape[2]):
        raise ValueError(f"q, k, v, g, beta must share the head count (got {q.shape[2]}, {k.shape[2]}, {v.shape[2]}, "
                         f"{g.shape[2]}, {beta.shape[2]}): grouped key/value heads are not supported by the kernels")
    if scale is None:
        scale = K ** -0.5                                                                 # chunk.py:373-374
    else:
        assert scale > 0, "Scale must be positive."
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    g = g.contiguous()
    if g.dtype != torch.float32:
        g = g.float()
    beta = beta.contiguous()
    if beta.dtype != q.dtype:
        beta = beta.to(q.dtype)
    if initial_state is not None:
        if initial_state.dtype not in _DT_CODE:
            initial_state = initial_state.float()
        initial_state = initial_state.contiguous()
        if tuple(initial_state.shape) != (NS, H, K, V):
            raise ValueError(f"initial_state shape {tuple(initial_state.shape)} != {(NS, H, K, V)}")
    ht = None
    if final_state_out is not None:
        if tuple(final_state_out.shape) != (NS, H, K, V) or final_state_out.dtype not in _DT_CODE \
                or not final_state_out.is_contiguous():
            raise ValueError("final_state_out must be a contiguous [B,H,K,V] fp32/bf16 tensor")
        ht = final_state_out
    elif output_final_state:
        ht = torch.empty(NS, H, K, V, dtype=torch.float32, device=q.device)              # chunk_delta_h.py:291
    o = torch.empty(B, T, H, V, dtype=q.dtype, device=q.device)
    return q, k, v, g, beta, float(scale), initial_state, ht, o, (B, T, H, K, V)


def _varlen_segments(cu_seqlens: torch.Tensor, T: int):
    """[(first, last)) token ranges of the flattened sequences (fla:ops/utils/index.py: the reference, too, reads the
    offsets on the host to lay out its chunk grid).  One host read of `cu_seqlens`: not capturable in a hipGraph."""
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("variable-length inputs read cu_seqlens on the host: not inside a hipGraph capture")
    cu = [int(x) for x in cu_seqlens.tolist()]
    if len(cu) < 2 or cu[0] != 0 or cu[-1] != T or any(b_ < a for a, b_ in zip(cu[:-1], cu[1:])):
        raise ValueError(f"cu_seqlens must rise from 0 to the flattened length {T} (got {cu[:4]}...{cu[-1:]})")
    return list(zip(cu[:-1], cu[1:]))


def _gdn_varlen(entry, q, k, v, g, beta, o, h0, ht, segs, H, K, V, scale, l2norm, extra):
    """Variable-length inputs (cu_seqlens; fla:ops/gated_delta_rule/chunk.py:355-369, chunk_delta_h.py:32-124): the sequences
    are independent -- own initial / final state, chunks counted from the sequence's own first token -- so each one is one call
    of the kernel on its slice of the flattened [1, T, ...] tensors (contiguous views; no copies).  InfiniteVL itself never
    takes this path (std:1223 nulls the mask)."""
    for i, (a, b_) in enumerate(segs):
        hi = None if h0 is None else h0[i:i + 1]
        ho = None if ht is None else ht[i:i + 1]
        if b_ == a:                                   # empty sequence: the state passes through
            if ho is not None:
                ho.zero_() if hi is None else ho.copy_(hi)
            continue
        entry(q[:, a:b_], k[:, a:b_], v[:, a:b_], g[:, a:b_], beta[:, a:b_], o[:, a:b_], hi, ho, b_ - a, H, K, V, scale, l2norm, *extra)


def _from_head_first(q, k, v, g, beta, cu_seqlens):
    """fla's deprecated [B,H,T,.] layout (chunk.py:361-373): rearranged to the time-major layout the kernels read."""
    if cu_seqlens is not None:
        raise RuntimeError("Sequences with variable lengths are not supported for head-first mode")
    q, k, v = (x.transpose(1, 2).contiguous() for x in (q, k, v))
    g, beta = (x.transpose(1, 2).contiguous() for x in (g, beta))
    return q, k, v, g, beta


def fused_recurrent_gated_delta_rule(
    q, k, v, g, beta, scale=None, initial_state=None, output_final_state=False, cu_seqlens=None,
    head_first=False, use_qk_l2norm_in_kernel=False, *, final_state_out: Optional[torch.Tensor] = None,
):
    """Token-recurrent gated delta rule (fla:ops/gated_delta_rule/fused_recurrent.py:218-335).

    q,k [B,T,H,K] bf16, v [B,T,H,V] bf16, g [B,T,H] fp32 log-decay, beta [B,T,H]; returns
    (o [B,T,H,V] bf16, final_state [B,H,K,V] fp32 or None).  `final_state_out` (extension) makes the
    kernel write the state straight into a caller tensor (fp32 or bf16, may alias initial_state)."""
    if head_first:
        q, k, v, g, beta = _from_head_first(q, k, v, g, beta, cu_seqlens)
    q, k, v, g, beta, scale, h0, ht, o, (B, T, H, K, V) = _gdn_common(
        q, k, v, g, beta, scale, initial_state, output_final_state, cu_seqlens, final_state_out)
    lib = _lib.load()

    rec_fwd = lib.ivl_gdn_recurrent_f16_fwd if q.dtype == torch.float16 else lib.ivl_gdn_recurrent_fwd

    def entry(q_, k_, v_, g_, b_, o_, hi, ho, Tn, H_, K_, V_, sc, l2):
        _lib.check(rec_fwd(
            _p(q_), _p(k_), _p(v_), _p(g_), _p(b_), _p(o_),
            _p(hi), _DT_CODE[hi.dtype] if hi is not None else IVL_F32,
            _p(ho), _DT_CODE[ho.dtype] if ho is not None else IVL_F32,
            q_.shape[0], Tn, H_, K_, V_, sc, l2, _stream(q_)))
    if cu_seqlens is not None:
        _gdn_varlen(entry, q, k, v, g, beta, o, h0, ht, _varlen_segments(cu_seqlens, T), H, K, V, scale,
                    int(bool(use_qk_l2norm_in_kernel)), ())
    else:
        entry(q, k, v, g, beta, o, h0, ht, T, H, K, V, scale, int(bool(use_qk_l2norm_in_kernel)))
    return (o.transpose(1, 2) if head_first else o), ht


def chunk_gated_delta_rule(
    q, k, v, g, beta, scale=None, initial_state=None, output_final_state=False, cu_seqlens=None,
    head_first=False, use_qk_l2norm_in_kernel=False, *, final_state_out: Optional[torch.Tensor] = None,
    mma_dtype=None,
):
    """Chunkwise gated delta rule, chunk 64 (fla:ops/gated_delta_rule/chunk.py:272-392).  Same I/O
    as fused_recurrent_gated_delta_rule; any T >= 1.  `mma_dtype="fp8_e4m3"` (extension, configs[4]) runs the serial
    pass's products on e4m3 operands."""
    if head_first:
        q, k, v, g, beta = _from_head_first(q, k, v, g, beta, cu_seqlens)
    q, k, v, g, beta, scale, h0, ht, o, (B, T, H, K, V) = _gdn_common(
        q, k, v, g, beta, scale, initial_state, output_final_state, cu_seqlens, final_state_out)
    if q.dtype == torch.float16:
        # IEEE-half activations (accepted by fla, chunk.py:352; never used by InfiniteVL): the MFMA kernels of the chunk path are
        # bf16 / e4m3 only, so the call runs on the token-recurrent kernel's fp16 instance -- the same function, fp32 arithmetic on
        # the fp16 inputs without the chunk form's intermediate roundings (within the operator's tolerance of the reference's fp16
        # chunk result: fixture gdn_chunk_T160_fp16)
        if mma_code(mma_dtype) != IVL_BF16:
            raise ValueError("mma_dtype applies to bf16 activations only")
        o, ht = fused_recurrent_gated_delta_rule(q, k, v, g, beta, scale=scale, initial_state=initial_state,
                                                 output_final_state=output_final_state, cu_seqlens=cu_seqlens,
                                                 use_qk_l2norm_in_kernel=use_qk_l2norm_in_kernel, final_state_out=final_state_out)
        return (o.transpose(1, 2) if head_first else o), ht
    lib = _lib.load()
    segs = _varlen_segments(cu_seqlens, T) if cu_seqlens is not None else None
    Tmax = max([b_ - a for a, b_ in segs] + [1]) if segs is not None else T
    nbytes = lib.ivl_gdn_chunk_workspace_bytes(B, Tmax, H, K, V)
    if nbytes == 0:
        raise ValueError(f"chunk_gated_delta_rule: unsupported head shape K={K}, V={V} (built for 128/256)")
    _, ws = _gdn_area_and_workspace(q.device, nbytes)          # per stream / per graph, like the flag words
    mma = mma_code(mma_dtype)

    def entry(q_, k_, v_, g_, b_, o_, hi, ho, Tn, H_, K_, V_, sc, l2):
        _lib.check(lib.ivl_gdn_chunk_fwd(
            _p(q_), _p(k_), _p(v_), _p(g_), _p(b_), _p(o_),
            _p(hi), _DT_CODE[hi.dtype] if hi is not None else IVL_F32,
            _p(ho), _DT_CODE[ho.dtype] if ho is not None else IVL_F32,
            q_.shape[0], Tn, H_, K_, V_, sc, l2, mma, _p(ws), ws.numel(), _stream(q_)))
    if segs is not None:
        _gdn_varlen(entry, q, k, v, g, beta, o, h0, ht, segs, H, K, V, scale, int(bool(use_qk_l2norm_in_kernel)), ())
    else:
        entry(q, k, v, g, beta, o, h0, ht, T, H, K, V, scale, int(bool(use_qk_l2norm_in_kernel)))
    return (o.transpose(1, 2) if head_first else o), ht


_GDN_SYNC: Dict[Tuple[int, object], torch.Tensor] = {}
_GDN_SYNC_OWNED = weakref.WeakValueDictionary()      # areas handed out by new_gdn_sync_area: they die with their owner (a GraphedStep)


def _all_gdn_sync_areas(dev: int):
    areas = [a for (d, _), a in list(_GDN_SYNC.items()) if d == dev]
    areas += [a for (d, _), a in list(_GDN_SYNC_OWNED.items()) if d == dev]
    return areas
_GDN_SYNC_SCOPE = threading.local()      # .areas: {device index: area} inside `with gdn_sync_scope(area)`
_GDN_SINGLE_LAUNCH = True      # tests switch it off to compare the single-launch form with the two-launch form


def new_gdn_sync_area(device) -> torch.Tensor:
    """A fresh, zeroed sync area for ivl_gdn_chunk_fused_fwd's single-launch forms (include/ivl_hip.h).  Whoever may issue
    calls CONCURRENTLY with others (a stream of its own, a captured hipGraph that is replayed beside other work) owns one."""
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("a GDN sync area cannot be created during hipGraph capture; create it (or run a warm-up step) first")
    area = torch.zeros(_lib.IVL_GDN_SYNC_BYTES, dtype=torch.uint8, device=device)
    _GDN_SYNC_OWNED[(area.device.index, id(area))] = area    # known to gdn_sync_check(deep=True) / gdn_sync_reset while its owner lives
    return area


@contextlib.contextmanager
def gdn_sync_scope(area: torch.Tensor):
    """Calls issued by this thread inside the scope use `area` (the harness captures every graph inside a scope with an area
    the graph object owns: graphs never share flag words with each other or with eager calls)."""
    areas = getattr(_GDN_SYNC_SCOPE, "areas", None)
    if areas is None:
        areas = _GDN_SYNC_SCOPE.areas = {}
    dev = area.device.index
    prev = areas.get(dev)
    areas[dev] = area
    try:
        yield area
    finally:
        if prev is None:
            areas.pop(dev, None)
        else:
            areas[dev] = prev


def _gdn_sync_area(device: torch.device) -> torch.Tensor:
    """The flag words of ivl_gdn_chunk_fused_fwd's single-launch forms: zeroed here once, then owned by the library (every
    launch leaves it all-zero).  Resolution: the enclosing gdn_sync_scope, else one area per (device, STREAM) for eager calls
    -- two calls that may run concurrently must not share flag words, and the package issues a call on the caller's current
    stream -- else, for a capture outside any scope, the device's one area for such graphs (they must not be replayed
    concurrently with each other; created together with the first eager area, i.e. by the warm-up a capture needs anyway)."""
    device = torch.device(device)
    dev = device.index if device.index is not None else torch.cuda.current_device()
    areas = getattr(_GDN_SYNC_SCOPE, "areas", None)
    if areas:
        area = areas.get(dev)
        if area is not None:
            return area
    capturing = torch.cuda.is_current_stream_capturing()
    key = (dev, "graphs") if capturing else (dev, int(torch.cuda.current_stream(device).cuda_stream))
    area = _GDN_SYNC.get(key)
    if area is None:
        if capturing:
            raise RuntimeError("the GDN sync area would have to be created during hipGraph capture; run a warm-up step first")
        area = torch.zeros(_lib.IVL_GDN_SYNC_BYTES, dtype=torch.uint8, device=device)
        _GDN_SYNC[key] = area
        if (dev, "graphs") not in _GDN_SYNC:
            _GDN_SYNC[(dev, "graphs")] = torch.zeros(_lib.IVL_GDN_SYNC_BYTES, dtype=torch.uint8, device=device)
    return area


_GDN_WS: Dict[int, list] = {}       # id(sync area) -> [records workspace, baked into a graph?, retired buffers]


def _gdn_workspace(nbytes: int, area: torch.Tensor) -> torch.Tensor:
    """The chunk-record workspace of the GDN calls that use sync area `area`.  Keyed like the sync area itself: two
    calls that may run concurrently -- two streams, a replayed graph beside eager calls, two graphs -- have their own flag words
    AND their own records; with one records buffer per device the pre-pass of one launch could overwrite records the scan of the
    other had already seen flagged (silently wrong outputs, no IVL_ERR_SYNC).  Lives as long as the area (a GraphedStep's
    workspace dies with it); growth rules as get_workspace: never during capture, a buffer baked into a graph is retired, not
    freed, for as long as the area lives."""
    key = id(area)
    ent = _GDN_WS.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None or ent[0].numel() < nbytes:
        if capturing:
            raise RuntimeError("the GDN workspace would have to grow during hipGraph capture; run a warm-up step first")
        retired = ent[2] if ent is not None else []
        grown = 0
        if ent is not None:
            if ent[1]:
                retired.append(ent[0])
            n0 = ent[0].numel()
            grown = 2 * n0 if n0 < (256 << 20) else n0 + n0 // 4
        else:
            weakref.finalize(area, _GDN_WS.pop, key, None)
        ent = [torch.empty(max(int(nbytes), grown, 1 << 20), dtype=torch.uint8, device=area.device), False, retired]
        _GDN_WS[key] = ent
    if capturing:
        ent[1] = True
    return ent[0]


_GDN_MIRROR_CAP = 16 << 20     # eager calls up to this size keep the scope-less-capture workspace of the device as large (see below)
_GDN_EAGER_NEED: Dict[int, int] = {}     # device index -> largest records workspace an eager scope-less call has asked for


def _gdn_area_and_workspace(device: torch.device, nbytes: int):
    """(sync area, records workspace) of a GDN chunk call on the current stream / scope.

    Footprint: one records workspace per sync area, i.e. per stream that has issued eager GDN chunk calls and per
    GraphedStep / gdn_sync_scope owner, each as large as the largest call it has seen (a 4096-token call at B = 1: ~64 MB), for the
    life of the area -- release_gdn_workspaces() drops the eager ones.  A capture OUTSIDE any scope uses the device's one area for
    such graphs and must find its workspace in place (nothing can be created while capturing): eager scope-less calls of up to
    16 MB of records (the step shapes) keep it as large as their own; larger calls only RECORD their need, and whoever captures
    such a call without a scope calls prepare_gdn_capture() first (bench.py's kernel timings do)."""
    area = _gdn_sync_area(device)
    ws = _gdn_workspace(nbytes, area)
    if not getattr(_GDN_SYNC_SCOPE, "areas", None) and not torch.cuda.is_current_stream_capturing():
        dev = area.device.index
        _GDN_EAGER_NEED[dev] = max(_GDN_EAGER_NEED.get(dev, 0), int(nbytes))
        garea = _GDN_SYNC.get((dev, "graphs"))
        if garea is not None and garea is not area and nbytes <= _GDN_MIRROR_CAP:
            _gdn_workspace(nbytes, garea)
    return area, ws


def prepare_gdn_capture(device=None) -> None:
    """Before capturing GDN chunk calls OUTSIDE a gdn_sync_scope (GraphedStep / GraphedDecode capture inside their own scope and do
    not need this): size the records workspace of the device's area for scope-less graphs to the largest eager call seen so far
    (the warm-up of the call that is about to be captured).  No-op when nothing is missing."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    dev = device.index if device.index is not None else torch.cuda.current_device()
    garea = _GDN_SYNC.get((dev, "graphs"))
    need = _GDN_EAGER_NEED.get(dev, 0)
    if garea is not None and need > 0:
        _gdn_workspace(need, garea)


def release_gdn_workspaces(device=None) -> None:
    """Drop the sync areas and records workspaces of EAGER calls on `device` (every stream's; the next eager call creates its own
    again).  Areas owned by a GraphedStep / a scope live and die with their owner.  Synchronises the device first."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    dev = device.index if device.index is not None else torch.cuda.current_device()
    torch.cuda.synchronize(device)
    for key in [k for k in _GDN_SYNC if k[0] == dev]:
        area = _GDN_SYNC.pop(key)
        _GDN_WS.pop(id(area), None)
    _GDN_EAGER_NEED.pop(dev, None)


def gdn_resident_blocks(override: Optional[int] = None) -> int:
    """Workgroups of the single-launch GDN kernels taken to be resident at once on the current device (what gates the
    single-launch forms: ivl_gdn_resident_blocks).  `override` None = a pure read (nothing changes); >= 0 replaces the
    occupancy-derived number process-wide (0 = always the two-launch form; the environment variable IVL_GDN_RESIDENT_BLOCKS
    sets it when the package is imported); < 0 restores the query.  Returns the number in force."""
    return _lib.load().ivl_gdn_resident_blocks(_lib.IVL_GDN_RESIDENT_QUERY if override is None else int(override))


def gdn_sync_check(device=None, *, deep: bool = False) -> None:
    """Raise IvlError(IVL_ERR_SYNC) if a wait inside a single-launch GDN call on `device` ran out (a broken contract: shared sync
    area, starved launch).  Free by default -- it reads the host-visible status word the kernels report through, so it sees
    failures of kernels that have FINISHED (call it behind a synchronisation point); `deep=True` also copies every sync area's
    own error word back (blocking).  The harness calls it after graph replays; every eager gdn_chunk_fused call checks it too
    (inside the C entry point)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    lib = _lib.load()
    with torch.cuda.device(device):
        _lib.check(lib.ivl_gdn_sync_status(None, None))
        if deep:
            dev = device.index if device.index is not None else torch.cuda.current_device()
            for area in _all_gdn_sync_areas(dev):
                _lib.check(lib.ivl_gdn_sync_status(_p(area), _stream(area)))


def gdn_sync_reset(device=None) -> None:
    """Re-arm every sync area of `device` after a reported failure (the outputs / states of the failed call are incomplete:
    the caller restarts from a known cache state)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    dev = device.index if device.index is not None else torch.cuda.current_device()
    lib = _lib.load()
    with torch.cuda.device(device):
        torch.cuda.synchronize(device)
        for area in _all_gdn_sync_areas(dev):
            _lib.check(lib.ivl_gdn_sync_reset(_p(area), _stream(area)))
        torch.cuda.synchronize(device)


def gdn_chunk_fused(proj: torch.Tensor, cols, conv_weights, conv_states_in, conv_states_out, A_log32, dt_bias32,
                    H: int, K: int, V: int, scale=None, initial_state=None, final_state_out=None, mma_dtype=None):
    """Chunked gated delta rule with its front end (3 short convs + SiLU + gate math) inside the pre-pass: one call from
    the fused projection output to the mixer core's output (std:1253-1323).  proj [B,T,ld] bf16; cols = (col_q, col_k,
    col_v, col_a, col_b); conv_weights 3 x [D,1,4] bf16; conv_states_in / _out 3 x ([B,D,4] bf16 or None; out may alias in).
    Returns o [B,T,H,V] bf16; the final recurrent state is written into `final_state_out` when given."""
    _need_gpu(proj)
    B, T, ld = proj.shape
    assert proj.is_contiguous() and proj.dtype == torch.bfloat16
    lib = _lib.load()
    nbytes = lib.ivl_gdn_chunk_workspace_bytes(B, T, H, K, V)
    if nbytes == 0:
        raise ValueError(f"gdn_chunk_fused: unsupported head shape K={K}, V={V} (built for 128/256)")
    area, ws = _gdn_area_and_workspace(proj.device, nbytes)    # records: one buffer per sync area (per stream / per graph)
    o = torch.empty(B, T, H, V, dtype=torch.bfloat16, device=proj.device)
    h0, ht = initial_state, final_state_out
    for s in (h0, ht):
        if s is not None:
            assert s.is_contiguous() and tuple(s.shape) == (B, H, K, V), (s.shape, (B, H, K, V))
    wq, wk, wv = conv_weights
    si, so = conv_states_in, conv_states_out
    _lib.check(lib.ivl_gdn_chunk_fused_fwd(
        _p(proj), ld, cols[0], cols[1], cols[2], cols[3], cols[4], _p(wq), _p(wk), _p(wv),
        _p(si[0]), _p(si[1]), _p(si[2]), _p(so[0]), _p(so[1]), _p(so[2]), _p(A_log32), _p(dt_bias32), _p(o),
        _p(h0), _DT_CODE[h0.dtype] if h0 is not None else IVL_F32, _p(ht), _DT_CODE[ht.dtype] if ht is not None else IVL_F32,
        B, T, H, K, V, wq.shape[-1], float(K ** -0.5 if scale is None else scale), mma_code(mma_dtype), _p(ws), ws.numel(),
        _p(area) if _GDN_SINGLE_LAUNCH else None, _stream(proj)))
    return o


def rope_tables(position_ids: torch.Tensor, inv_freq: torch.Tensor, attention_scaling: float = 1.0):
    """cos / sin tables bf16 [..., 2 * len(inv_freq)] of int64 position ids (any leading shape) in one launch (std:896-930)."""
    _need_gpu(position_ids, inv_freq)
    pos = position_ids if position_ids.dtype == torch.int64 and position_ids.is_contiguous() else position_ids.to(torch.int64).contiguous()
    half = inv_freq.numel()
    cos = torch.empty(*pos.shape, 2 * half, dtype=torch.bfloat16, device=pos.device)
    sin = torch.empty_like(cos)
    _lib.check(_lib.load().ivl_rope_tables_fwd(_p(pos), _p(inv_freq), _p(cos), _p(sin), pos.numel(), half,
                                               float(attention_scaling), _stream(pos)))
    return cos, sin


# Host-side check of cu_seqlens / max_seqlen in vision_window_attention (one device sync per eager call; never under capture).
_VALIDATE_VISION_SEGMENTS = False


def vision_window_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int,
                            scaling: Optional[float] = None, rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """Non-causal attention inside each segment [cu_seqlens[s], cu_seqlens[s+1]) of a packed patch sequence: the
    per-window attention_interface loop / flash-attention varlen call of InfiniteVLVisionAttention.forward
    (strm:752-796), with apply_rotary_pos_emb_vision (strm:657-671) folded in when `rope=(cos, sin)` ([S, d] fp32) is given.

    q, k, v: [S, H, d] bf16, any token / head strides with a contiguous channel dimension (the slices of the fused qkv
    projection are taken as they are).  cu_seqlens: int32 [n_seg + 1] on the device, never read on the host;
    `max_seqlen` bounds the segment lengths.  Returns [S, H, d] bf16 contiguous (what `proj` consumes after a reshape)."""
    _need_gpu(q, k, v, cu_seqlens)
    S, H, d = q.shape
    assert k.shape == q.shape and v.shape == q.shape, "q, k, v must share [S, H, d]"
    assert q.dtype == k.dtype == v.dtype == torch.bfloat16, "bf16 activations"
    fix = lambda x: x if x.stride(-1) == 1 and x.stride(0) % 8 == 0 and x.stride(1) % 8 == 0 and x.data_ptr() % 16 == 0 else x.contiguous()
    q, k, v = fix(q), fix(k), fix(v)
    cu = cu_seqlens if cu_seqlens.dtype == torch.int32 and cu_seqlens.is_contiguous() else cu_seqlens.to(torch.int32).contiguous()
    cos = sin = None
    if rope is not None:
        cos, sin = (x if x.dtype == torch.float32 and x.is_contiguous() else x.float().contiguous() for x in rope)
        assert tuple(cos.shape) == (S, d) and tuple(sin.shape) == (S, d), "vision rotary tables are [S, head_dim]"
    # tokens outside every segment (cu_seqlens[-1] < S) and tiles beyond an under-estimated max_seqlen are skipped by the
    # kernel: they read as zeros, never as uninitialised memory; outside a graph capture the bound itself is checked
    o = torch.zeros(S, H, d, dtype=torch.bfloat16, device=q.device)
    if not torch.cuda.is_current_stream_capturing() and _VALIDATE_VISION_SEGMENTS:
        lens = cu[1:] - cu[:-1]
        if int(cu[-1]) != S or (lens.numel() and int(lens.max()) > int(max_seqlen)):
            raise ValueError(f"vision_window_attention: cu_seqlens must cover all {S} tokens (ends at {int(cu[-1])}) and "
                             f"max_seqlen={int(max_seqlen)} must bound the segment lengths (max {int(lens.max())})")
    lib = _lib.load()
    ws_bytes = lib.ivl_vision_attn_workspace_bytes(S, H, d, int(max_seqlen)) if cos is not None else 0
    ws = get_workspace(ws_bytes, q.device, "vision") if ws_bytes else None
    _lib.check(lib.ivl_vision_attn_fwd(
        _p(q), _p(k), _p(v), _p(o), q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
        o.stride(0), o.stride(1), _p(cu), cu.numel() - 1, int(max_seqlen), S, H, d,
        float(d ** -0.5 if scaling is None else scaling), _p(cos), _p(sin), _p(ws), ws.numel() if ws is not None else 0, _stream(q)))
    return o


def gdn_gate(a: torch.Tensor, b: torch.Tensor, A_log: torch.Tensor, dt_bias: torch.Tensor):
    """g = -exp(A_log) * softplus(a + dt_bias) (fp32), beta = sigmoid(b) (bf16); std:1293-1294.
    a, b are the a_proj / b_proj outputs [..., H] in bf16."""
    _need_gpu(a, b)
    H = a.shape[-1]
    a, b = a.contiguous(), b.contiguous()
    A32 = A_log.detach().float().contiguous()
    dt32 = dt_bias.detach().float().contiguous()
    g = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    beta = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device)
    _lib.check(_lib.load().ivl_gdn_gate_fwd(_p(a), _p(b), _p(A32), _p(dt32), _p(g), _p(beta),
                                            a.numel() // H, H, _stream(a)))
    return g, beta


class ShortConvolution(nn.Module):
    """Causal depthwise conv1d (+SiLU) with a [N,D,W] raw-input cache, newest last
    (fla:modules/convolution.py:128-297).  Parameter name/shape match the checkpoint:
    `weight` [D,1,W].  A given cache is updated IN PLACE and returned.  For (cache given, T>1) the
    cached inputs are carried in (pip fla 0.4.0 / streaming semantics)."""

    def __init__(self, hidden_size: int, kernel_size: int, bias: bool = False, activation: Optional[str] = "silu",
                 use_fast_conv1d: Optional[bool] = True, device=None, dtype=None, **_unused):
        super().__init__()
        if activation is not None:
            assert activation in ["silu", "swish"], f"Activation `{activation}` not supported yet."
        self.hidden_size = hidden_size
        self.kernel_size = (kernel_size,)
        self.activation = activation
        self.weight = nn.Parameter(torch.empty(hidden_size, 1, kernel_size, device=device, dtype=dtype))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:                                                         # nn.Conv1d's parameter and initialisation (convolution.py:128-160)
            self.bias = nn.Parameter(torch.empty(hidden_size, device=device, dtype=dtype))
            bound = 1.0 / (kernel_size ** 0.5)                           # fan_in = (in_channels / groups) * kernel_size
            nn.init.uniform_(self.bias, -bound, bound)
        else:
            self.register_parameter("bias", None)

    def extra_repr(self) -> str:
        return (f"{self.hidden_size}, {self.hidden_size}, kernel_size={self.kernel_size}, groups={self.hidden_size}, "
                f"bias={self.bias is not None}, activation={self.activation}")

    @property
    def state_size(self) -> int:                                         # convolution.py:295-297
        return self.hidden_size * self.kernel_size[0]

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[torch.Tensor] = None,
                output_final_state: bool = False, cu_seqlens: Optional[torch.Tensor] = None, **kwargs):
        _need_gpu(x)
        if cu_seqlens is not None:
            # flattened variable-length batch (convolution.py:216-231): x [1, T, D], cache [N, D, W]; the sequences do not see
            # each other's tokens -- one call per sequence on its slice (InfiniteVL never takes this path: std:1223)
            if x.shape[0] != 1:
                raise ValueError(f"The batch size is expected to be 1 rather than {x.shape[0]} when using `cu_seqlens`.")
            if mask is not None:                                         # convolution.py:226-228
                raise ValueError("`mask` and `cu_seqlens` cannot be provided at the same time")
            segs = _varlen_segments(cu_seqlens, x.shape[1])
            if x.dtype != torch.bfloat16:
                raise ValueError(f"ShortConvolution kernel is built for bf16, got {x.dtype}")
            x = x.contiguous()
            D, W = x.shape[2], self.kernel_size[0]
            if cache is not None and (cache.dtype != torch.bfloat16 or not cache.is_contiguous()
                                      or tuple(cache.shape) != (len(segs), D, W)):
                raise ValueError(f"conv cache must be a contiguous bf16 [N,D,W] = {(len(segs), D, W)} tensor (one row block per "
                                 f"sequence of cu_seqlens); got {cache.dtype} {tuple(cache.shape)}")
            state_in = cache
            if cache is None and output_final_state:
                cache = torch.zeros(len(segs), D, W, dtype=x.dtype, device=x.device)
            y = torch.empty_like(x)
            for i, (a, b_) in enumerate(segs):
                if b_ > a:
                    self._launch(x[:, a:b_], y[:, a:b_], None if state_in is None else state_in[i:i + 1],
                                 None if cache is None else cache[i:i + 1], 1, b_ - a, D, W)
            return y, cache
        if mask is not None:
            x = x.mul(mask.unsqueeze(-1))
        if x.dtype != torch.bfloat16:
            raise ValueError(f"ShortConvolution kernel is built for bf16, got {x.dtype}")
        B, T, D = x.shape
        W = self.kernel_size[0]
        x = x.contiguous()
        state_in = cache
        if output_final_state and cache is None:
            cache = torch.empty(B, D, W, dtype=x.dtype, device=x.device)       # zero history, written by the kernel
        if cache is not None and (cache.dtype != torch.bfloat16 or not cache.is_contiguous()
                                  or tuple(cache.shape) != (B, D, W)):
            raise ValueError("conv cache must be a contiguous bf16 [N,D,W] tensor")
        y = torch.empty_like(x)
        self._launch(x, y, state_in, cache, B, T, D, W)
        return y, cache

    def _launch(self, x, y, state_in, state_out, B, T, D, W) -> None:
        w = self.weight
        if w.dtype != torch.bfloat16:
            w = w.to(torch.bfloat16)
        if self.bias is not None:
            _lib.check(_lib.load().ivl_short_conv_bias_fwd(
                _p(x), _p(w.contiguous()), _p(self.bias.to(torch.bfloat16).contiguous()), _p(state_in), _p(y), _p(state_out), B, T, D, W,
                int(self.activation is not None), _stream(x)))
        else:
            _lib.check(_lib.load().ivl_short_conv_fwd(
                _p(x), _p(w.contiguous()), _p(state_in), _p(y), _p(state_out), B, T, D, W,
                int(self.activation is not None), _stream(x)))

    def step(self, x: torch.Tensor, cache: torch.Tensor):
        return self.forward(x, cache=cache, output_final_state=True)


class FusedRMSNormGated(nn.Module):
    """y = rmsnorm(x) * weight * g * sigmoid(g) over the last dim (256)
    (fla:modules/fused_norm_gate.py:735-796)."""

    def __init__(self, hidden_size: int, elementwise_affine: bool = True, eps: float = 1e-5,
                 activation: str = "swish", device=None, dtype=None):
        super().__init__()
        if activation not in ("swish", "silu"):
            raise ValueError(f"Unsupported activation: {activation}")
        self.hidden_size, self.elementwise_affine, self.eps, self.activation = hidden_size, elementwise_affine, eps, activation
        if elementwise_affine:
            self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        else:
            self.register_parameter("weight", None)
        self.register_parameter("bias", None)

    def forward(self, x: torch.Tensor, g: torch.Tensor, residual=None, prenorm=False, residual_in_fp32=False):
        _need_gpu(x, g, residual)
        if x.dtype != torch.bfloat16:
            raise ValueError(f"FusedRMSNormGated kernel is built for bf16, got {x.dtype}")
        N = x.shape[-1]
        x, g = x.contiguous(), g.contiguous()
        y = torch.empty_like(x)
        if self.weight is None:                                         # elementwise_affine=False: x * rstd * 1
            w = torch.ones(N, dtype=torch.bfloat16, device=x.device)
        else:
            w = self.weight if self.weight.dtype == torch.bfloat16 else self.weight.to(torch.bfloat16)
        if residual is not None or prenorm or residual_in_fp32:
            # fla's residual path (fused_norm_gate.py:98-155, 688-721): row = x + residual in fp32; residual_out in the residual's
            # dtype (fp32 when only residual_in_fp32 is set); returned when prenorm
            if residual is not None:
                if residual.dtype not in _DT_CODE or tuple(residual.shape) != tuple(x.shape):
                    raise ValueError("residual must be a bf16 / fp32 tensor of x's shape")
                residual = residual.contiguous()
            res_dt = residual.dtype if residual is not None else (torch.float32 if residual_in_fp32 else None)
            res_out = torch.empty(x.shape, dtype=res_dt, device=x.device) if res_dt is not None and (residual is not None or res_dt != x.dtype) else None
            _lib.check(_lib.load().ivl_rmsnorm_swish_gate_res_fwd(
                _p(x), _p(g), _p(w.contiguous()), _p(residual), _DT_CODE[residual.dtype] if residual is not None else IVL_BF16,
                _p(res_out), _DT_CODE[res_out.dtype] if res_out is not None else IVL_BF16, _p(y), x.numel() // N, N, float(self.eps), _stream(x)))
            return y if not prenorm else (y, res_out if res_out is not None else x)
        _lib.check(_lib.load().ivl_rmsnorm_swish_gate_fwd(_p(x), _p(g), _p(w.contiguous()), _p(y),
                                                          x.numel() // N, N, float(self.eps), _stream(x)))
        return y


class RMSNorm(nn.Module):
    """y = rmsnorm(x) * weight over the last dim (fla.modules.RMSNorm; the reference's `o_norm` when
    `use_gate=False`, std:1213 / std:1341).  Same constructor and parameter names as fla's.  At the width InfiniteVL uses it
    for (head_v_dim = 256) the arithmetic is fla's own (fla:modules/layernorm.py: x * rstd * w in fp32, rounded ONCE):
    ivl_rmsnorm_swish_gate_fwd without a gate.  Other widths go through ivl_add_rmsnorm_fwd, which rounds x * rstd to bf16
    before the weight (the Qwen2RMSNorm form of the decoder layers): up to 1 bf16 ulp from fla there."""

    def __init__(self, hidden_size: int, elementwise_affine: bool = True, bias: bool = False, eps: float = 1e-5,
                 device=None, dtype=None):
        super().__init__()
        if bias:
            raise NotImplementedError("InfiniteVL's norms carry no bias")
        self.hidden_size, self.elementwise_affine, self.eps = hidden_size, elementwise_affine, eps
        self.register_parameter("weight", None)
        self.register_parameter("bias", None)
        if elementwise_affine:
            self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))

    def extra_repr(self) -> str:
        return f"{self.hidden_size}, eps={self.eps}" + ("" if self.elementwise_affine else ", elementwise_affine=False")

    def forward(self, x: torch.Tensor, residual=None, prenorm: bool = False, residual_in_fp32: bool = False):
        if residual is not None or prenorm:
            raise NotImplementedError("residual/prenorm are not used by InfiniteVL (std:1341)")
        w = self.weight if self.weight is not None else torch.ones(self.hidden_size, dtype=x.dtype, device=x.device)
        if self.hidden_size == 256 and x.dtype == torch.bfloat16:
            _need_gpu(x)
            xc = x.contiguous()
            y = torch.empty_like(xc)
            _lib.check(_lib.load().ivl_rmsnorm_swish_gate_fwd(_p(xc), None, _p(w.to(torch.bfloat16).contiguous()), _p(y),
                                                              xc.numel() // 256, 256, float(self.eps), _stream(xc)))
            return y
        y, _ = add_rmsnorm(x, None, w, self.eps)
        return y


def get_unpad_data(attention_mask: torch.Tensor):
    """(indices, cu_seqlens, max_seqlen_in_batch) of a 0/1 padding mask [B, S] (fla.layers.utils; std:1254).  The GDN
    mixer nulls its mask (std:1223), so nothing on the hot path reaches this; kept so that `std:52` is a pure rename."""
    lens = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    cu = torch.nn.functional.pad(torch.cumsum(lens, dim=0, dtype=torch.int32), (1, 0))
    return indices, cu, int(lens.max().item()) if lens.numel() else 0


def index_first_axis(x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    """Rows `indices` of x [(b s), ...] (fla.layers.utils; std:1255-1258)."""
    return x.index_select(0, indices)


def pad_input(x: torch.Tensor, indices: torch.Tensor, batch: int, seqlen: int) -> torch.Tensor:
    """Inverse of index_first_axis: x [total, ...] scattered into zeros [batch, seqlen, ...] (std:1345)."""
    out = torch.zeros(batch * seqlen, *x.shape[1:], dtype=x.dtype, device=x.device)
    out.index_copy_(0, indices, x)
    return out.view(batch, seqlen, *x.shape[1:])


# ---------------------------------------------------------------------------------------------
# fused prologue / epilogue
# ---------------------------------------------------------------------------------------------
def gdn_prologue(proj: torch.Tensor, cols, conv_weights, conv_states_in, conv_states_out, A_log32, dt_bias32,
                 H: int, Dq: int, Dk: int, Dv: int):
    """3 short convs (+SiLU, carry-in) + gate math from ONE fused projection output.
    proj [B,T,ld] bf16 (contiguous rows); cols = (col_q, col_k, col_v, col_a, col_b); conv_weights 3 x [D,1,4]
    bf16; conv_states_in 3 x ([B,D,4] bf16 or None); conv_states_out likewise (may alias the inputs).
    Returns q [B,T,Dq], k [B,T,Dk], v [B,T,Dv] bf16, g fp32 [B,T,H], beta bf16 [B,T,H]."""
    _need_gpu(proj)
    B, T, ld = proj.shape
    assert proj.is_contiguous() and proj.dtype == torch.bfloat16
    dev = proj.device
    q = torch.empty(B, T, Dq, dtype=torch.bfloat16, device=dev)
    k = torch.empty(B, T, Dk, dtype=torch.bfloat16, device=dev)
    v = torch.empty(B, T, Dv, dtype=torch.bfloat16, device=dev)
    g = torch.empty(B, T, H, dtype=torch.float32, device=dev)
    beta = torch.empty(B, T, H, dtype=torch.bfloat16, device=dev)
    wq, wk, wv = conv_weights
    si, so = conv_states_in, conv_states_out
    _lib.check(_lib.load().ivl_gdn_prologue_fwd(
        _p(proj), ld, cols[0], cols[1], cols[2], cols[3], cols[4], _p(wq), _p(wk), _p(wv),
        _p(si[0]), _p(si[1]), _p(si[2]), _p(so[0]), _p(so[1]), _p(so[2]), _p(A_log32), _p(dt_bias32),
        _p(q), _p(k), _p(v), _p(g), _p(beta), B, T, H, Dq, Dk, Dv, wq.shape[-1], 1, _stream(proj)))
    return q, k, v, g, beta


def gdn_decode_step(proj: torch.Tensor, cols, conv_weights, conv_states, A_log32, dt_bias32, norm_weight, eps: float,
                    state: torch.Tensor, H: int, K: int, V: int, scale: float) -> torch.Tensor:
    """One new token per sequence through the GDN mixer core in ONE launch (convs + gates + delta rule + gated
    norm).  proj [B,1,ld] bf16; cols = (col_q, col_k, col_v, col_g, col_a, col_b); conv_states 3 x [B,D,4] bf16 and
    state [B,H,K,V] are updated in place.  Returns y [B,1,H*V] bf16 (the o_proj input)."""
    _need_gpu(proj, state)
    B, T, ld = proj.shape
    assert T == 1 and proj.is_contiguous() and proj.dtype == torch.bfloat16 and state.is_contiguous()
    y = torch.empty(B, 1, H * V, dtype=torch.bfloat16, device=proj.device)
    wq, wk, wv = conv_weights
    sq, sk, sv = conv_states
    _lib.check(_lib.load().ivl_gdn_decode_step_fwd(
        _p(proj), ld, cols[0], cols[1], cols[2], cols[3], cols[4], cols[5], _p(wq), _p(wk), _p(wv), _p(sq), _p(sk), _p(sv),
        _p(A_log32), _p(dt_bias32), _p(norm_weight), float(eps), _p(state), _DT_CODE[state.dtype], _p(y), B, H, K, V,
        float(scale), _stream(proj)))
    return y


# The GDN decode step on 64 workgroups + gated norm / conv-state shift in the o_proj launch (gdn_decode_split + gdn_out_linear):
# bit-identical to the one-launch step + plain o_proj, built and measured in round 5 -- 1.953 vs 1.925 ms per token
# at 8K context (tools/ab_decode_split.py, ABAB in one process): it LOSES, so the modules keep the one-launch step.
_SPLIT_DECODE = False


def gdn_decode_split(proj: torch.Tensor, cols, conv_weights, conv_states, A_log32, dt_bias32, state: torch.Tensor,
                     H: int, K: int, V: int, scale: float) -> torch.Tensor:
    """The GDN decode step on 4 x as many workgroups (one per sequence, head and quarter of the value columns): returns the delta
    rule's UN-NORMALISED bf16 output [B,1,H*V]; the v conv state and the recurrent state are updated in place, the q / k conv
    states are only read -- gdn_out_linear (the o_proj launch) applies the gated norm and shifts them.  cols = (col_q, col_k,
    col_v, col_a, col_b)."""
    _need_gpu(proj, state)
    B, T, ld = proj.shape
    assert T == 1 and proj.is_contiguous() and proj.dtype == torch.bfloat16 and state.is_contiguous()
    o_raw = torch.empty(B, 1, H * V, dtype=torch.bfloat16, device=proj.device)
    wq, wk, wv = conv_weights
    sq, sk, sv = conv_states
    _lib.check(_lib.load().ivl_gdn_decode_split_fwd(
        _p(proj), ld, cols[0], cols[1], cols[2], cols[3], cols[4], _p(wq), _p(wk), _p(wv), _p(sq), _p(sk), _p(sv),
        _p(A_log32), _p(dt_bias32), _p(state), _DT_CODE[state.dtype], _p(o_raw), B, H, K, V, float(scale), _stream(proj)))
    return o_raw


def gdn_out_linear(o_raw: torch.Tensor, proj: torch.Tensor, col_g: int, col_q: int, col_k: int, norm_weight: torch.Tensor, eps: float,
                   conv_state_q: torch.Tensor, conv_state_k: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                   H: int) -> torch.Tensor:
    """o_proj(o_norm(o_raw, gate)) for a decode step, one launch: the gated RMSNorm runs in the prologue of the weight stream, and
    the launch shifts the q / k conv states (the second half of gdn_decode_split's state update).  proj [B,1,ld] is the step's
    fused projection (gate columns at col_g, raw q / k columns at col_q / col_k)."""
    _need_gpu(o_raw, proj, weight)
    B, T, ld = proj.shape
    K = o_raw.shape[-1]
    N = weight.shape[0]
    assert T == 1 and o_raw.is_contiguous() and weight.is_contiguous() and weight.dtype == torch.bfloat16
    y = torch.empty(B, 1, N, dtype=torch.bfloat16, device=o_raw.device)
    gate = proj.view(B, ld)[:, col_g:]
    _lib.check(_lib.load().ivl_gdn_out_linear_small_m_fwd(
        _p(o_raw), ctypes.c_void_p(gate.data_ptr()), ld, _p(norm_weight), float(eps), H, _p(proj), ld, col_q, col_k,
        _p(conv_state_q), _p(conv_state_k), _p(weight), _p(bias) if bias is not None else None, _p(y), B, N, K, _stream(o_raw)))
    return y


def rmsnorm_swish_gate_strided(x: torch.Tensor, gate_base: torch.Tensor, gate_ld: int, weight: torch.Tensor,
                               eps: float) -> torch.Tensor:
    """Gated RMSNorm with the gate read in place from a fused projection buffer.  x [B,T,H,256] bf16
    contiguous; gate_base = view whose data_ptr is the gate block's first element, row stride gate_ld."""
    _need_gpu(x, gate_base)
    B, T, H, N = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.load().ivl_rmsnorm_swish_gate_strided_fwd(_p(x), _p(gate_base), gate_ld, H, _p(weight), _p(y),
                                                              B * T * H, N, float(eps), _stream(x)))
    return y


def add_rmsnorm(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, eps: float):
    """h = x + residual (bf16) ; y = RMSNorm(h) * weight, one launch.  Returns (y, h); with residual=None, h is x."""
    _need_gpu(x, residual)
    if x.dtype != torch.bfloat16:
        raise ValueError("add_rmsnorm is built for bf16")
    N = x.shape[-1]
    x = x.contiguous()
    y = torch.empty_like(x)
    h = torch.empty_like(x) if residual is not None else None
    w = weight if weight.dtype == torch.bfloat16 else weight.to(torch.bfloat16)
    _lib.check(_lib.load().ivl_add_rmsnorm_fwd(_p(x), _p(residual.contiguous()) if residual is not None else None,
                                               _p(w), _p(y), _p(h), x.numel() // N, N, float(eps), _stream(x)))
    return y, (h if h is not None else x)


_PRENORM = True            # tests switch it off to compare the fused norm + projection launches with the separate ones


class PreNorm:
    """A (residual add +) RMSNorm that has NOT been launched: the decoder stack hands it to a mixer / MLP in place of the
    normalised input during a decode step (<= 4 rows), and `linear` / `linear_swiglu` run it in the prologue of their
    weight-stream kernel (ivl_norm_linear_small_m_fwd: same arithmetic as ivl_add_rmsnorm_fwd, one launch less per norm --
    72 per token).  `h` is the new residual stream x + residual (x itself without a residual): allocated here, written by
    whichever kernel ends up running the norm.  Anything else that needs the normalised tensor calls materialize()."""

    def __init__(self, x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, eps: float):
        self.x = x if x.is_contiguous() else x.contiguous()
        self.residual = None if residual is None else (residual if residual.is_contiguous() else residual.contiguous())
        self.weight = weight if weight.dtype == torch.bfloat16 else weight.to(torch.bfloat16)
        self.eps = float(eps)
        self.h = torch.empty_like(self.x) if residual is not None else self.x
        self.shape, self.dtype, self.device, self.is_cuda = self.x.shape, self.x.dtype, self.x.device, self.x.is_cuda
        self._y = None
        self.done = False                      # the norm has run (h is valid)

    def size(self, *a):
        return self.x.size(*a)

    def dim(self):
        return self.x.dim()

    def numel(self):
        return self.x.numel()

    def materialize(self) -> torch.Tensor:
        if self._y is None:
            N = self.x.shape[-1]
            self._y = torch.empty_like(self.x)
            _lib.check(_lib.load().ivl_add_rmsnorm_fwd(_p(self.x), _p(self.residual), _p(self.weight), _p(self._y),
                                                       _p(self.h) if self.residual is not None else None,
                                                       self.x.numel() // N, N, self.eps, _stream(self.x)))
            self.done = True
        return self._y


def _prenorm_linear(pn: "PreNorm", weight: torch.Tensor, bias: Optional[torch.Tensor], glu: bool) -> Optional[torch.Tensor]:
    """The fused launch when it applies (decode step, bf16, K <= 4096), else None."""
    K = weight.shape[-1]
    rows = pn.numel() // K if K else 0
    if not (pn.is_cuda and 1 <= rows <= 4 and pn.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.dim() == 2
            and weight.is_contiguous() and K % 8 == 0 and 512 <= K <= 4096 and pn.shape[-1] == K and pn._y is None
            and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous()))):
        return None
    N = weight.shape[0] // 2 if glu else weight.shape[0]
    y = torch.empty(*pn.shape[:-1], N, dtype=torch.bfloat16, device=pn.device)
    _lib.check(_lib.load().ivl_norm_linear_small_m_fwd(
        _p(pn.x), _p(pn.residual), _p(pn.weight), pn.eps, _p(pn.h) if pn.residual is not None else None,
        _p(weight), _p(bias) if bias is not None else None, _p(y), rows, N, K, 1 if glu else 0, _stream(pn.x)))
    pn.done = True
    return y


def silu_mul(gate_up: torch.Tensor) -> torch.Tensor:
    """SwiGLU gate on a fused gate|up projection [..., 2I] -> [..., I]."""
    _need_gpu(gate_up)
    assert gate_up.is_contiguous() and gate_up.dtype == torch.bfloat16
    I = gate_up.shape[-1] // 2
    y = torch.empty(*gate_up.shape[:-1], I, dtype=torch.bfloat16, device=gate_up.device)
    _lib.check(_lib.load().ivl_silu_mul_fwd(_p(gate_up), _p(y), gate_up.numel() // (2 * I), I, _stream(gate_up)))
    return y


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear forward.  A single-token decode step (<= 4 rows) is a pure weight stream and goes through
    ivl_linear_small_m_fwd; longer calls are stock library GEMMs (hipBLASLt / rocBLAS through torch), which
    SURVEY.md 8(d) keeps outside the hot path."""
    if isinstance(x, PreNorm):
        y = _prenorm_linear(x, weight, bias, False)
        if y is not None:
            return y
        x = x.materialize()
    K = weight.shape[-1]
    rows = x.numel() // K if K else 0
    if (x.is_cuda and 1 <= rows <= 4 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and weight.dim() == 2 and weight.is_contiguous() and K % 8 == 0
            and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous()))):
        xc = x if x.is_contiguous() else x.contiguous()
        N = weight.shape[0]
        y = torch.empty(*x.shape[:-1], N, dtype=torch.bfloat16, device=x.device)
        _lib.check(_lib.load().ivl_linear_small_m_fwd(_p(xc), _p(weight), _p(bias) if bias is not None else None,
                                                      _p(y), rows, N, K, _stream(x)))
        return y
    return torch.nn.functional.linear(x, weight, bias)


def linear_swiglu(x: torch.Tensor, w_gate_up: torch.Tensor) -> torch.Tensor:
    """silu(gate_proj(x)) * up_proj(x) with the fused gate|up weight [2I, K] (std:945).  Decode steps (<= 4 rows)
    apply the gate in the epilogue of the weight-stream kernel; longer calls are a library GEMM + silu_mul."""
    if isinstance(x, PreNorm):
        y = _prenorm_linear(x, w_gate_up, None, True)
        if y is not None:
            return y
        x = x.materialize()
    K = w_gate_up.shape[-1]
    I = w_gate_up.shape[0] // 2
    rows = x.numel() // K
    if (x.is_cuda and 1 <= rows <= 4 and x.dtype == torch.bfloat16 and w_gate_up.dtype == torch.bfloat16
            and w_gate_up.is_contiguous() and K % 8 == 0):
        xc = x if x.is_contiguous() else x.contiguous()
        y = torch.empty(*x.shape[:-1], I, dtype=torch.bfloat16, device=x.device)
        _lib.check(_lib.load().ivl_linear_swiglu_small_m_fwd(_p(xc), _p(w_gate_up), None, _p(y), rows, I, K, _stream(x)))
        return y
    return silu_mul(linear(x, w_gate_up))


def apply_mrope_strided_inplace(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, mrope_section):
    """M-RoPE in place on q [B,T,Hq,d] / k [B,T,Hkv,d] VIEWS into a fused qkv projection (row = token)."""
    _need_gpu(q, k, cos, sin)
    B, T, Hq, d = q.shape
    Hkv = k.shape[2]
    assert q.stride(3) == 1 and q.stride(2) == d and k.stride(3) == 1 and k.stride(2) == d

    def row_stride(t_):      # elements between consecutive tokens of the flattened [B*T] row index
        if T > 1 and B > 1:
            assert t_.stride(0) == T * t_.stride(1), "q/k must be row-uniform views of a [B*T, ld] buffer"
        return t_.stride(1) if T > 1 else t_.stride(0)
    cos = cos.to(torch.bfloat16).contiguous()
    sin = sin.to(torch.bfloat16).contiguous()
    s0, s1, s2 = (int(s) for s in mrope_section)
    _lib.check(_lib.load().ivl_mrope_strided_fwd(_p(q), _p(k), row_stride(q), row_stride(k), _p(cos), _p(sin),
                                                 B, T, Hq, Hkv, d, s0, s1, s2, _stream(q)))
    return q, k


# ---------------------------------------------------------------------------------------------
# sliding-window attention
# ---------------------------------------------------------------------------------------------
def apply_mrope_inplace(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, mrope_section):
    """M-RoPE on time-major q [B,T,Hq,d] / k [B,T,Hkv,d] in place; cos/sin [3,B,T,d] bf16
    (std:949-984).  Bit-identical to the reference's eager bf16 arithmetic."""
    _need_gpu(q, k, cos, sin)
    B, T, Hq, d = q.shape
    Hkv = k.shape[2]
    assert q.is_contiguous() and k.is_contiguous() and q.dtype == k.dtype == torch.bfloat16
    cos = cos.to(torch.bfloat16).contiguous()
    sin = sin.to(torch.bfloat16).contiguous()
    if cos.shape[0] != 3 or tuple(cos.shape[1:]) != (B, T, d):
        raise ValueError(f"cos/sin must be [3,B,T,d]; got {tuple(cos.shape)}")
    s0, s1, s2 = (int(s) for s in mrope_section)
    _lib.check(_lib.load().ivl_mrope_fwd(_p(q), _p(k), _p(cos), _p(sin), B, T, Hq, Hkv, d, s0, s1, s2, _stream(q)))
    return q, k


# the 256-row kernel for long calls over a full ring (swa_ring256.hip); IVL_SWA_RING256=0 keeps every call on the 128-row kernel
SWA_RING256 = os.environ.get("IVL_SWA_RING256", "1") != "0"
SWA_RING256_CALLS = 0          # calls that took it (tests read this to know which kernel a product-path call ran on)


def swa_forward(
    q: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor, *, window: Optional[int], scaling: float,
    k_cache: Optional[torch.Tensor] = None, v_cache: Optional[torch.Tensor] = None,
    pos: int = 0, pos_dev: Optional[torch.Tensor] = None, n_query: Optional[int] = None,
    layout: str = "bthd", mma_dtype=None, rope=None, append: bool = False, pos_min: int = 0, pos_min_holds_in_graph: bool = False,
) -> torch.Tensor:
    """Sliding-window GQA attention over (ring cache ++ new keys); returns o [B,T,Hq,d] bf16.

    layout "bthd": q [B,T,Hq,d], k_new/v_new [B,T_new,Hkv,d]; "bhtd": head-major views
    (any strides with a contiguous last dim are accepted -- no copies are made).
    `n_query` = T (defaults to q's length); T_new >= T, the first T_new-T new keys being older keys.
    `rope=(cos, sin, mrope_section)` (cos/sin bf16 [3,B,T,d]): q and k_new are the UN-rotated projections and M-RoPE
    (std:949-984) is applied while they are loaded -- bit-identical to apply_mrope_inplace followed by this call.
    `append=True`: the call's tokens are also appended to the ring afterwards (swa_cache_append's work, folded into the
    split-KV combine launch when there is one).
    `pos_min`: a lower bound of the position the caller vouches for (the host-side counter of the cache) when the position
    itself is read from `pos_dev`: a long call over a FULL ring (pos_min >= C) takes the 256-row kernel on a linear copy of the
    keys (ivl_swa_args.pos_min).  Ignored while a stream capture is recording (a replay may start from an earlier position)
    unless `pos_min_holds_in_graph` says the bound holds for every replay (the kernel micro-benchmarks replay one position)."""
    _need_gpu(q, k_new, v_new, k_cache, v_cache, pos_dev)
    if q.dtype != torch.bfloat16 or k_new.dtype != torch.bfloat16 or v_new.dtype != torch.bfloat16:
        raise ValueError("swa_forward is built for bf16")
    if layout == "bhtd":
        q, k_new, v_new = q.transpose(1, 2), k_new.transpose(1, 2), v_new.transpose(1, 2)
    B, T, Hq, d = q.shape
    T_new, Hkv = k_new.shape[1], k_new.shape[2]
    if n_query is not None:
        assert n_query == T
    if q.stride(-1) != 1:
        q = q.contiguous()
    if k_new.stride(-1) != 1 or v_new.stride() != k_new.stride():
        k_new, v_new = k_new.contiguous(), v_new.contiguous()
    C = 0
    if k_cache is not None:
        assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.dtype == torch.bfloat16
        assert tuple(k_cache.shape[:2]) == (B, Hkv) and k_cache.shape[3] == d
        C = k_cache.shape[2]
    o = torch.empty(B, T, Hq, d, dtype=torch.bfloat16, device=q.device)
    lib = _lib.load()
    nbytes = lib.ivl_swa_workspace_bytes(B, T, Hq, d)
    pos_min = int(pos_min) if pos_dev is not None else int(pos)
    if not (SWA_RING256 and C > 0 and pos_min >= C and T >= 256 and T_new == T and window == C + 1) or (
            torch.cuda.is_current_stream_capturing() and not pos_min_holds_in_graph):
        pos_min = 0
    else:
        n256 = lib.ivl_swa_ring256_workspace_bytes(B, T, Hq, Hkv, d, C)      # 0: the shape does not qualify (the launcher decides the same way)
        if n256 > 0 and mma_code(mma_dtype) == IVL_BF16:
            global SWA_RING256_CALLS
            SWA_RING256_CALLS += 1
        nbytes = max(nbytes, n256)
    ws = get_workspace(nbytes, q.device, "swa")
    a = SwaArgs()
    a.q, a.k_new, a.v_new, a.k_cache, a.v_cache, a.o = (q.data_ptr(), k_new.data_ptr(), v_new.data_ptr(),
                                                         k_cache.data_ptr() if C else None,
                                                         v_cache.data_ptr() if C else None, o.data_ptr())
    a.q_sb, a.q_st, a.q_sh = q.stride(0), q.stride(1), q.stride(2)
    a.kn_sb, a.kn_st, a.kn_sh = k_new.stride(0), k_new.stride(1), k_new.stride(2)
    a.B, a.T, a.T_new, a.Hq, a.Hkv, a.d = B, T, T_new, Hq, Hkv, d
    a.cache_capacity = C
    a.window = int(window) if window is not None else 0
    a.pos = int(pos)
    a.pos_dev = pos_dev.data_ptr() if pos_dev is not None else None
    a.scaling = float(scaling)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    a.mma_dtype = mma_code(mma_dtype)
    a.append_new = int(bool(append) and C > 0)
    a.pos_min = pos_min
    if rope is not None:
        cos, sin, sec = _rope_args(rope, B, T, d)
        a.rope_cos, a.rope_sin, a.rope_s0, a.rope_s1 = cos.data_ptr(), sin.data_ptr(), int(sec[0]), int(sec[1])
    _lib.check(lib.ivl_swa_fwd(ctypes.byref(a), _stream(q)))
    return o


def _rope_args(rope, B: int, T: int, d: int):
    cos, sin, sec = rope
    if cos.dtype != torch.bfloat16 or not cos.is_contiguous():
        cos = cos.to(torch.bfloat16).contiguous()
    if sin.dtype != torch.bfloat16 or not sin.is_contiguous():
        sin = sin.to(torch.bfloat16).contiguous()
    if tuple(cos.shape) != (3, B, T, d) or tuple(sin.shape) != (3, B, T, d):
        raise ValueError(f"rope cos/sin must be [3,B,T,d] = {(3, B, T, d)}; got {tuple(cos.shape)}")
    return cos, sin, sec


def swa_cache_append(k_new: torch.Tensor, v_new: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                     pos: int = 0, pos_dev: Optional[torch.Tensor] = None, rope=None) -> None:
    """Write the call's new tokens [B,T,Hkv,d] into the ring (slot (pos+t) % C); std:146-172.  `rope` as in
    swa_forward: k_new is un-rotated and is rotated on the way into the ring."""
    _need_gpu(k_new, v_new, k_cache, v_cache)
    B, T, Hkv, d = k_new.shape
    if k_new.stride(-1) != 1 or v_new.stride() != k_new.stride():
        k_new, v_new = k_new.contiguous(), v_new.contiguous()
    rc = rs = None
    s0 = s1 = 0
    if rope is not None:
        rc, rs, sec = _rope_args(rope, B, T, d)
        s0, s1 = int(sec[0]), int(sec[1])
    _lib.check(_lib.load().ivl_swa_cache_append(
        _p(k_new), _p(v_new), k_new.stride(0), k_new.stride(1), k_new.stride(2), _p(k_cache), _p(v_cache),
        B, T, Hkv, d, k_cache.shape[2], int(pos), _p(pos_dev), _p(rc), _p(rs), s0, s1, _stream(k_new)))


def counter_add(counter: torch.Tensor, delta: int) -> None:
    _lib.check(_lib.load().ivl_counter_add(_p(counter), int(delta), _stream(counter)))


def swa_attention_interface(module, query, key, value, attention_mask=None, dropout: float = 0.0,
                            scaling: Optional[float] = None, sliding_window: Optional[int] = None,
                            position_ids=None, **kwargs):
    """Drop-in for ALL_ATTENTION_FUNCTIONS["flash_attention_2"] at std:1097-1108:
    query [B,Hq,T,d], key/value [B,Hkv,S,d] (S >= T, already concatenated with the cached keys);
    causal, bottom-right aligned, left window `sliding_window`; the mask tensor is ignored exactly as
    FlashAttention-2 ignores it for an unpadded batch.  Returns (out [B,T,Hq,d], None)."""
    if dropout:
        raise NotImplementedError("attention dropout is a training feature; the hot path is inference")
    if scaling is None:
        scaling = query.shape[-1] ** -0.5
    out = swa_forward(query, key, value, window=sliding_window, scaling=scaling, layout="bhtd")
    return out, None
