"""Multi-GPU side of the path: sequences are independent, so the batch is
sharded over ranks (one process per GPU, weights replicated, all recurrent state rank-local) and the
ONLY exchange is one all-gather of the last-position logits [B_local, vocab] over RCCL/xGMI at the
very end -- latency-bound (~300 KB per rank), no per-layer collective, no ring all-reduce.

Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2)."""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend: str = "nccl") -> Tuple[int, int, int]:
    """Initialise torch.distributed when WORLD_SIZE > 1 (MASTER_ADDR/PORT from the env; 127.0.0.1 on
    one node).  Keeps HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only supports dmabuf IPC."""
    rank, world, local_rank = env_world()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 and not dist.is_initialized():
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
