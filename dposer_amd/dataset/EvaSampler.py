"""Contiguous-shard evaluation sampler -- the only data-parallel partitioning logic of the reference
(lib/dataset/EvaSampler.py:6-119): rank r gets a contiguous block, the first ``total % G`` ranks get
one extra element, nothing is padded or dropped."""
import torch
from torch.utils.data import Sampler

from ..distributed import shard_bounds


class DistributedEvalSampler(Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=False, seed=0):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("Requires distributed package to be available")
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.epoch, self.shuffle, self.seed = 0, shuffle, seed
        self.total_size = len(dataset)
        lo, hi = shard_bounds(self.total_size, num_replicas, rank)
        self.num_samples = hi - lo

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            indices = torch.randperm(self.total_size, generator=g).tolist()
        else:
            indices = list(range(self.total_size))
        lo, hi = shard_bounds(self.total_size, self.num_replicas, self.rank)
        return iter(indices[lo:hi])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
