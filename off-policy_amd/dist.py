"""Data-parallel plumbing: one process per GPU, gradients exchanged in ONE flat all-reduce (RCCL over xGMI).

The reference has no distributed training (its only collective, `average_gradients`, offpolicy/utils/util.py:148-153,
is dead code). The update shards over episodes: each rank back-propagates the UN-normalised loss sum of its share
of the sampled episodes, then a single SUM all-reduce of `[grads | loss_sum | mask_count | qtot_sum | 0]`
(~475 KB for QMIX 3s5z) makes every rank hold the global sums; normalisation by the global mask count, global-norm
clipping and Adam then run redundantly and identically on every rank.
"""
import numpy as np
import torch


def is_distributed():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def world():
    if is_distributed():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def shard_indices(inds, rank=None, world_size=None):
    """Rank r's contiguous share of the sampled episode indices (B must divide evenly)."""
    if rank is None:
        rank, world_size = world()
    inds = np.asarray(inds)
    assert len(inds) % world_size == 0, "batch size must be a multiple of the number of ranks"
    per = len(inds) // world_size
    return inds[rank * per:(rank + 1) * per]


def allreduce_flat_(flat, group=None):
    """In-place SUM all-reduce of one flat tensor; a single collective per training step."""
    if is_distributed():
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=group)
    return flat
