"""Multi-GPU side of the path: sequences are independent, so the batch is
sharded over ranks (one process per GPU, weights replicated, all recurrent state rank-local) and the
ONLY exchange is one all-gather of the last-position logits [B_local, vocab] over RCCL/xGMI at the
very end -- latency-bound (~300 KB per rank), no per-layer collective, no ring all-reduce.

sequence_parallel_prefill() is the one extension beyond the reference: ONE long
sequence is cut into `world` consecutive segments; rank r runs the whole layer stack on segment r, and for every
decoder layer receives the layer's carried state (GDN: conv + recurrent state, ~1 MiB; SWA: the W-1 key/value ring,
<= 8.4 MB) from rank r-1 right before its mixer and forwards its own to rank r+1 right after it -- point-to-point
over xGMI, no collective.  Rank r starts layer i as soon as rank r-1 has finished layer i, so the ranks form a
wavefront over (segment, layer) and a prefill of S segments x L layers takes (S + L - 1) layer-times instead of S*L.

Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2)."""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend: str = "nccl", force: bool = False) -> Tuple[int, int, int]:
    """Initialise torch.distributed when WORLD_SIZE > 1 (MASTER_ADDR/PORT from the env; 127.0.0.1 on
    one node).  Keeps HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only supports dmabuf IPC.  `force=True` creates the
    process group for a single process as well (the one-GPU RCCL contact test: librccl loads, the `device_id=` init path works)."""
    rank, world, local_rank = env_world()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_batch(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of `global_batch` sequences: returns (first_index, count) of this rank.
    The first (global_batch % world) ranks take one extra sequence."""
    if global_batch < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad shard request: batch={global_batch} rank={rank} world={world}")
    base, rem = divmod(global_batch, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device: torch.device) -> float:
    """MAX-reduce a host scalar (the timing contract: the slowest rank defines the step time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def describe_ranks(device: torch.device) -> dict:
    """What the process group actually runs on: backend, world size and one record per rank (rank, local device index, PCI
    bus id, device name) collected with all_gather_object -- so that a multi-GPU bench line proves where its ranks sat."""
    rec = {"rank": dist.get_rank(), "device_index": device.index, "pci_bus_id": None, "name": None}
    if device.type == "cuda":
        pr = torch.cuda.get_device_properties(device)
        rec["name"] = pr.name
        ids = [getattr(pr, a, None) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        rec["pci_bus_id"] = (":".join(f"{int(x):02x}" for x in ids) if all(x is not None for x in ids)
                             else str(getattr(pr, "uuid", device.index)))
    recs = [None] * dist.get_world_size()
    dist.all_gather_object(recs, rec)
    return {"backend": dist.get_backend(), "world": dist.get_world_size(), "devices": recs,
            "distinct_devices": len({r["pci_bus_id"] for r in recs})}


def gather_last_logits(logits_local: torch.Tensor, counts: List[int]) -> torch.Tensor:
    """All-gather last-position logits [B_local, V] from every rank into [sum(counts), V] (rank order
    = batch order of shard_batch).  Ragged shards are padded to max(counts) for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return logits_local
    world = dist.get_world_size()
    assert len(counts) == world
    bmax = max(counts)
    V = logits_local.shape[-1]
    padded = logits_local.new_zeros(bmax, V)
    padded[: logits_local.shape[0]] = logits_local
    out = logits_local.new_empty(world * bmax, V)
    dist.all_gather_into_tensor(out, padded.contiguous())
    out = out.view(world, bmax, V)
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)


# ---------------------------------------------------------------------------------------------------------
# sequence-parallel prefill: per-layer state hand-off between consecutive ranks
# ---------------------------------------------------------------------------------------------------------
def _p2p_stage_through_host(t: torch.Tensor) -> bool:
    """RCCL moves device tensors directly (xGMI peer-to-peer); gloo's send/recv take host tensors only."""
    return t.is_cuda and dist.get_backend() != "nccl"


def send_tensors(tensors: Sequence[torch.Tensor], dst: int) -> None:
    for t in tensors:
        dist.send(t.cpu() if _p2p_stage_through_host(t) else t, dst)


def recv_tensors_(tensors: Sequence[torch.Tensor], src: int) -> None:
    """Receive IN PLACE (tensor addresses stay: a captured hipGraph over the cache remains valid)."""
    for t in tensors:
        if _p2p_stage_through_host(t):
            buf = torch.empty(t.shape, dtype=t.dtype, device="cpu")
            dist.recv(buf, src)
            t.copy_(buf)
        else:
            dist.recv(t, src)


class _Pending:
    """One layer's hand-off in flight: ONE batch_isend_irecv for all its tensors (a single RCCL group call per hop
    instead of one blocking send/recv per tensor)."""

    def __init__(self, tensors: Sequence[torch.Tensor], peer: int, recv: bool):
        self.tensors, self.recv = list(tensors), recv
        self.stage = [torch.empty(t.shape, dtype=t.dtype, device="cpu") if recv else t.cpu()
                      for t in self.tensors] if any(_p2p_stage_through_host(t) for t in self.tensors) else None
        bufs = self.stage if self.stage is not None else self.tensors
        op = dist.irecv if recv else dist.isend
        self.reqs = dist.batch_isend_irecv([dist.P2POp(op, b_, peer) for b_ in bufs]) if bufs else []

    def wait(self) -> None:
        for r in self.reqs:
            r.wait()
        if self.recv and self.stage is not None:
            for t, b_ in zip(self.tensors, self.stage):
                t.copy_(b_)


def segment_bounds(total_tokens: int, rank: int, world: int, multiple: int = 64) -> Tuple[int, int]:
    """[first, last) token range of segment `rank`: equal segments rounded up to `multiple` tokens (the GDN chunk
    length, so that a segment boundary is a chunk boundary); trailing ranks may get a short or empty segment."""
    if total_tokens < 0 or world <= 0 or not (0 <= rank < world) or multiple <= 0:
        raise ValueError(f"bad segment request: tokens={total_tokens} rank={rank} world={world}")
    per = -(-total_tokens // world)
    per = -(-per // multiple) * multiple
    first = min(rank * per, total_tokens)
    return first, min(first + per, total_tokens)


def sequence_parallel_prefill(model: Callable, inputs_embeds: torch.Tensor, cache, first_token: int,
                              rank: Optional[int] = None, world: Optional[int] = None, logits_to_keep: int = 0):
    """Run `model` (an InfiniteVLTextStack, or anything with its forward(..., layer_hooks=) contract) on THIS rank's
    segment `inputs_embeds` [B, T_r, hidden] of one long sequence whose first token has absolute index `first_token`.
    `cache` is this rank's own pre-allocated cache; its layers are overwritten, layer by layer, with the state rank
    r-1 reached at the end of its segment, and handed on after this rank's segment.  Ranks with an empty segment
    (T_r == 0) just relay.  Returns model's (hidden, logits) for the local segment (None, None when empty).
    After the call the LAST rank's cache holds the state of the whole sequence."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    B, T = inputs_embeds.shape[0], inputs_embeds.shape[1]
    has_prev, has_next = rank > 0 and first_token > 0, rank + 1 < world
    n_layers = len(cache.layers)
    recvs, sends = {}, []

    def post_recv(i: int) -> None:
        if has_prev and i < n_layers and i not in recvs:
            recvs[i] = _Pending(cache.layers[i].carried_tensors(), rank - 1, recv=True)

    def before(i: int) -> None:
        # layer i's state was requested right behind layer i-1's send (its tensors are not touched until now), so the transfer
        # runs under layer i-1's MLP and whatever the sender still has to do before its mixer i
        if has_prev:
            post_recv(i)
            recvs.pop(i).wait()
            cache.layers[i].import_carried(first_token)

    def after(i: int) -> None:
        if has_next:                              # non-blocking: the MLP of this layer overlaps the transfer
            sends.append(_Pending(cache.layers[i].carried_tensors(), rank + 1, recv=False))
        # the next layer's receive goes out BEHIND this layer's send: RCCL runs the batches of a communicator in issue order,
        # and a receive posted first would hold the send back until rank r-1 has finished the NEXT layer (two layer-times per
        # hop of the wavefront instead of one)
        post_recv(i + 1)

    if T == 0:                                   # nothing to compute: pass every layer's state through
        for i in range(n_layers):
            before(i)
            after(i)
        out = (None, None)
    else:
        pos = torch.arange(first_token, first_token + T, device=inputs_embeds.device)[None, None, :].expand(3, B, T)
        out = model(inputs_embeds=inputs_embeds, position_ids=pos.contiguous(), past_key_values=cache,
                    logits_to_keep=logits_to_keep, layer_hooks=(before, after))
    for p_ in sends:
        p_.wait()
    return out
