"""Multi-GPU: clips are independent units (eval-mode BatchNorm has no cross-clip state; the
reference is bit-identical for B=1 vs B=2), so a global batch is sharded
contiguously over ranks with replicated weights and NO data-path collective; the only exchange is one
all-gather of the per-clip logits (RCCL over xGMI when the backend is "nccl"; 4 bytes per clip, purely
latency bound).  One process per GPU, launched with torch.distributed.run."""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str = None, timeout_s: float = 300.0) -> Tuple[int, int, int]:
    """Initialises torch.distributed from the torchrun environment (no-op for a single process).
    With the RCCL backend the process group is bound to this rank's GPU (``device_id``), so the communicator is created
    eagerly on the right device and barriers need no device guess; ``timeout_s`` bounds the rendezvous and every
    collective (a rank that died before joining makes the others fail within that time instead of hanging)."""
    import datetime
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, local_rank, world


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split; the first (global_batch % world) ranks take one extra clip."""
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_clips(clips: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(clips.shape[0], rank, world)
    return clips[lo:hi]


def gather_logits(local_logits: torch.Tensor, global_batch: int, out: torch.Tensor = None) -> torch.Tensor:
    """All ranks receive the (global_batch, K) logits in clip order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_logits
    world, rank = dist.get_world_size(), dist.get_rank()
    k = local_logits.shape[1]
    if global_batch % world == 0:                       # even shards: one all_gather_into_tensor
        if out is None:
            out = torch.empty((global_batch, k), dtype=local_logits.dtype, device=local_logits.device)
        dist.all_gather_into_tensor(out, local_logits.contiguous())
        return out
    # ragged shards: pad every rank's block to the largest shard, gather, trim (collectives need equal sizes)
    counts = [shard_bounds(global_batch, r, world)[1] - shard_bounds(global_batch, r, world)[0] for r in range(world)]
    cmax = max(counts)
    padded = torch.zeros((cmax, k), dtype=local_logits.dtype, device=local_logits.device)
    padded[:local_logits.shape[0]] = local_logits
    allp = torch.empty((world * cmax, k), dtype=local_logits.dtype, device=local_logits.device)
    dist.all_gather_into_tensor(allp, padded)
    return torch.cat([allp[r * cmax:r * cmax + counts[r]] for r in range(world)], 0)
