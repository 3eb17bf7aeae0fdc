"""Aspect-ratio group sampling for data-parallel training (mmdet/datasets/samplers/group_sampler.py:60-148).

Every mini-batch of `samples_per_gpu` images comes from ONE group (`dataset.flag`: 1 for landscape, 0 for portrait
images), so a batch pads to one orientation; every rank sees the same number of batches.  `DistributedGroupSampler` is
deterministic in (epoch, world size): the same `torch.Generator` sequence as the reference, hence the same indices
(tests/test_data_pipeline.py compares with the reference's class)."""
import math

import numpy as np
import torch
from torch.utils.data import Sampler


class DistributedGroupSampler(Sampler):

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            num_replicas = num_replicas if num_replicas is not None else (dist.get_world_size() if on else 1)
            rank = rank if rank is not None else (dist.get_rank() if on else 0)
        assert hasattr(dataset, 'flag')
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.num_replicas, self.rank, self.epoch = num_replicas, rank, 0
        self.flag = np.asarray(dataset.flag)
        self.group_sizes = np.bincount(self.flag)
        chunk = samples_per_gpu * num_replicas
        # per group: rounded up to a whole number of (batch x replicas); per replica that is ceil(size/chunk) batches
        self.num_samples = sum(int(math.ceil(s / chunk)) * samples_per_gpu for s in self.group_sizes)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        chunk = self.samples_per_gpu * self.num_replicas
        order = []
        for gid, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            members = np.where(self.flag == gid)[0]
            members = members[torch.randperm(int(size), generator=g).numpy()].tolist()
            padded = int(math.ceil(size / chunk)) * chunk
            reps, rest = divmod(padded - size, size)
            order += members * (1 + reps) + members[:rest]        # pad by repeating the shuffled group
        assert len(order) == self.total_size
        batches = torch.randperm(len(order) // self.samples_per_gpu, generator=g).tolist()
        order = [order[j] for b in batches for j in range(b * self.samples_per_gpu, (b + 1) * self.samples_per_gpu)]
        start = self.num_samples * self.rank
        return iter(order[start:start + self.num_samples])


class GroupSampler(Sampler):
    """Single-process variant (group_sampler.py:10-57): every group shuffled with `np.random`, padded to a whole
    number of mini-batches by random re-draws, mini-batches then permuted.  Same `np.random` call sequence as the
    reference, so a seeded run yields the same order."""

    def __init__(self, dataset, samples_per_gpu=1):
        assert hasattr(dataset, 'flag')
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.flag = np.asarray(dataset.flag).astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(np.ceil(s / samples_per_gpu)) * samples_per_gpu for s in self.group_sizes)

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        spg, parts = self.samples_per_gpu, []
        for gid, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            members = np.where(self.flag == gid)[0]
            np.random.shuffle(members)
            extra = int(np.ceil(size / spg)) * spg - len(members)
            parts.append(np.concatenate([members, np.random.choice(members, extra)]))
        flat = np.concatenate(parts)
        flat = np.concatenate([flat[b * spg:(b + 1) * spg] for b in np.random.permutation(range(len(flat) // spg))])
        assert len(flat) == self.num_samples
        return iter(flat.astype(np.int64).tolist())
