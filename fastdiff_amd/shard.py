"""Multi-GPU sharding of independent utterances.

The reference shards inference the same way: one process per GPU (utils/trainer.py:94-107), rank r takes the
utterances `r::world` of a DistributedSampler(shuffle=False) (tasks/vocoder/vocoder_base.py:43-49) and there is no
collective on the data path.  Here the partition is length-balanced (longest-processing-time greedy) because padded
micro-batches cost max(T_i); the optional scatter/gather moves mels out and waveforms back point-to-point over
torch.distributed (RCCL over xGMI on the GPUs, gloo in the CPU tests) -- messages are <1 MB per utterance, there is
no all-reduce anywhere.
"""
from typing import List, Sequence

import torch
import torch.distributed as dist


def partition_utterances(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy LPT: utterance indices per rank with ~equal total frames.  Deterministic; every index appears once."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(lengths[i])
    return [sorted(p) for p in parts]


def round_robin_partition(n_items: int, world_size: int) -> List[List[int]]:
    """The reference's own assignment: DistributedSampler(shuffle=False) gives rank r the indices r::world."""
    return [list(range(r, n_items, world_size)) for r in range(world_size)]


def micro_batches(indices: Sequence[int], lengths: Sequence[int], max_batch: int) -> List[List[int]]:
    """Bucket a rank's utterances (longest first) into padded micro-batches of at most `max_batch`."""
    order = sorted(indices, key=lambda i: (-int(lengths[i]), i))
    return [order[k:k + max_batch] for k in range(0, len(order), max_batch)]


def pad_mels(mels: Sequence[torch.Tensor], pad_value: float = 0.0) -> torch.Tensor:
    """[80,T_i] tensors -> zero-padded [B,80,max T] (collate_2d semantics, utils/__init__.py:136-150)."""
    T = max(m.shape[-1] for m in mels)
    out = mels[0].new_full((len(mels), mels[0].shape[0], T), pad_value)
    for i, m in enumerate(mels):
        out[i, :, :m.shape[-1]] = m
    return out


def scatter_utterances(mels: Sequence[torch.Tensor], parts: List[List[int]], src: int = 0, device=None):
    """Rank `src` holds all mels ([80,T_i] each); afterwards every rank holds its own (index, mel) list.
    Point-to-point isend/irecv; lengths travel first as one small broadcast."""
    rank, world = dist.get_rank(), dist.get_world_size()
    n = sum(len(p) for p in parts)
    lens = torch.zeros(n, dtype=torch.int64, device=device)
    if rank == src:
        lens = torch.tensor([m.shape[-1] for m in mels], dtype=torch.int64, device=device)
    dist.broadcast(lens, src=src)
    mine = []
    if rank == src:
        reqs = []
        for r in range(world):
            for i in parts[r]:
                if r == src:
                    mine.append((i, mels[i].to(device) if device is not None else mels[i]))
                else:
                    reqs.append(dist.isend(mels[i].contiguous().to(device) if device is not None else mels[i].contiguous(), dst=r, tag=i))
        for q in reqs:
            q.wait()
    else:
        for i in parts[rank]:
            buf = torch.empty((80, int(lens[i])), dtype=torch.float32, device=device)
            dist.recv(buf, src=src, tag=i)
            mine.append((i, buf))
    return mine, lens.tolist()


def gather_waveforms(mine, lens: Sequence[int], parts: List[List[int]], hop: int = 256, dst: int = 0, device=None):
    """Inverse of scatter_utterances: rank `dst` ends up with the list of waveforms ([T_i*hop] each) in index order."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank != dst:
        for i, wav in mine:
            dist.send(wav.contiguous(), dst=dst, tag=i)
        return None
    out = [None] * len(lens)
    for i, wav in mine:
        out[i] = wav
    for r in range(world):
        if r == dst:
            continue
        for i in parts[r]:
            buf = torch.empty(int(lens[i]) * hop, dtype=torch.float32, device=device)
            dist.recv(buf, src=r, tag=i)
            out[i] = buf
    return out
