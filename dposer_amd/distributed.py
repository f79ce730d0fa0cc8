"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The reference has no DDP training (run/train.py is single device) and shards evaluation with a
gloo process group (run/completion.py:83-92, lib/dataset/EvaSampler.py).  On an 8 x MI355X node the
training step is data parallel over the batch: every rank computes the DSM gradient of its
contiguous B/G shard into ONE flat fp32 buffer (8.28 M floats = 33.1 MB), which is summed with a
single all-reduce -- xGMI is a full mesh (7 links x ~153 GB/s per GPU), so one large message keeps
all seven links busy; bucketing would only add launches (there is nothing left to overlap with: the
whole backward is a dozen GEMM launches that finish together).  The 1/world factor and the
global-norm clip are folded into the fused Adam/EMA kernel (grad_scale).  Sampling, completion and
FK shard the batch with no collective at all.
"""
import os

import torch
import torch.distributed as dist


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_initialized() else 1


def rank():
    return dist.get_rank() if is_initialized() else 0


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_*).  Returns (rank, world_size, local_rank)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not is_initialized():
        if backend is None:
            backend = os.environ.get("DPOSER_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(lr % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rk, world_size=ws)
    return rk, ws, lr


def all_reduce_sum_(flat: torch.Tensor) -> int:
    """In-place SUM all-reduce of the flat gradient; returns the world size (the caller divides)."""
    if not is_initialized() or dist.get_world_size() == 1:
        return 1
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return dist.get_world_size()


def broadcast_(flat: torch.Tensor, src=0):
    """Make every rank start from rank ``src``'s parameters."""
    if is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def shard_bounds(total: int, num_replicas: int, rank_: int):
    """Contiguous split used by the reference's DistributedEvalSampler (lib/dataset/EvaSampler.py:78-84):
    base = total // G, the first total % G ranks get one extra element."""
    base, extra = divmod(total, num_replicas)
    start = base * rank_ + min(rank_, extra)
    return start, start + base + (1 if rank_ < extra else 0)


def barrier():
    if is_initialized():
        dist.barrier()
