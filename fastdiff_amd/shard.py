"""Multi-GPU sharding of independent utterances.

The reference shards inference the same way: one process per GPU (utils/trainer.py:94-107), rank r takes the
utterances `r::world` of a DistributedSampler(shuffle=False) (tasks/vocoder/vocoder_base.py:43-49) and there is no
collective on the data path.  Here the partition is length-balanced (longest-processing-time greedy) because padded
micro-batches cost max(T_i); the optional scatter/gather moves mels out and waveforms back point-to-point over
torch.distributed (RCCL over xGMI on the GPUs, gloo in the CPU tests): one packed message per peer each way, posted as one
grouped send/recv (dist.batch_isend_irecv) -- <= 2.2 MB out and 3.5 MB back per peer for BASELINE config 4; there is no
all-reduce anywhere.
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


def _as_bytes(t: torch.Tensor) -> torch.Tensor:
    """A flat uint8 view of a contiguous tensor: the one element type every backend moves (RCCL has no int16)."""
    return t.contiguous().view(-1).view(torch.uint8)


def _exchange(ops):
    """Post all point-to-point operations of this rank as ONE group (ncclGroupStart/End on RCCL, plain isend/irecv on gloo)
    and wait for them.  Messages between a pair of ranks are matched in posting order; no tags (RCCL ignores them).
    gloo moves host memory only: device buffers handed to it (the device-staged path exercised on a box without RCCL peers, e.g.
    two ranks sharing one GPU in the tests) go through host copies here, so the callers stay the same for both backends."""
    if not ops:
        return
    if dist.get_backend(ops[0].group) == "gloo" and any(op.tensor.is_cuda for op in ops):
        host = [op.tensor.cpu() if op.op is dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype) for op in ops]
        for req in dist.batch_isend_irecv([dist.P2POp(op.op, h, op.peer, op.group, op.tag) for op, h in zip(ops, host)]):
            req.wait()
        for op, h in zip(ops, host):
            if op.op is dist.irecv:
                op.tensor.copy_(h)
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def scatter_utterances(mels: Sequence[torch.Tensor], parts: List[List[int]], src: int = 0, device=None, lens: Sequence[int] = None):
    """Rank `src` holds all mels ([80,T_i] each); afterwards every rank holds its own (index, mel) list.
    Lengths travel first as one small broadcast (skipped when the caller already distributed them: `lens`); then ONE packed
    message per peer (its utterances back to back, in the order of parts[r]) in a single grouped send/recv -- at most world-1
    messages leave `src`, whatever the number of utterances.  Returns (own (index, mel) list, all lengths)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    n = sum(len(p) for p in parts)
    if lens is None:
        lens_t = torch.zeros(n, dtype=torch.int64, device=device)
        if rank == src:
            lens_t = torch.tensor([m.shape[-1] for m in mels], dtype=torch.int64, device=device)
        dist.broadcast(lens_t, src=src)
        lens = lens_t.tolist()
    lens_l = [int(v) for v in lens]
    mine = []
    if rank == src:
        ops, keep = [], []
        for r in range(world):
            if r == src:
                mine = [(i, mels[i].to(device) if device is not None else mels[i]) for i in parts[r]]
            elif parts[r]:
                buf = torch.cat([mels[i].reshape(-1).float() for i in parts[r]])
                buf = buf.to(device) if device is not None else buf.contiguous()
                keep.append(buf)
                ops.append(dist.P2POp(dist.isend, buf, r))
        _exchange(ops)
    elif parts[rank]:
        buf = torch.empty(80 * sum(lens_l[i] for i in parts[rank]), dtype=torch.float32, device=device)
        _exchange([dist.P2POp(dist.irecv, buf, src)])
        off = 0
        for i in parts[rank]:
            mine.append((i, buf[off: off + 80 * lens_l[i]].view(80, lens_l[i])))
            off += 80 * lens_l[i]
    return mine, lens_l


def gather_waveforms(mine, lens: Sequence[int], parts: List[List[int]], hop: int = 256, dst: int = 0, device=None,
                     dtype: torch.dtype = torch.float32):
    """Inverse of scatter_utterances: rank `dst` ends up with the list of waveforms ([T_i*hop] each, float32 or the int16 PCM of
    the device epilogue) in index order.  One packed message per peer, all receives of `dst` posted as one group."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank != dst:
        if mine:
            by_idx = dict(mine)
            buf = torch.cat([by_idx[i].reshape(-1).to(dtype) for i in parts[rank]])
            _exchange([dist.P2POp(dist.isend, _as_bytes(buf), dst)])
        return None
    out = [None] * len(lens)
    for i, wav in mine:
        out[i] = wav
    ops, bufs = [], {}
    for r in range(world):
        if r == dst or not parts[r]:
            continue
        bufs[r] = torch.empty(sum(int(lens[i]) * hop for i in parts[r]), dtype=dtype, device=device)
        ops.append(dist.P2POp(dist.irecv, _as_bytes(bufs[r]), r))
    _exchange(ops)
    for r, buf in bufs.items():
        off = 0
        for i in parts[r]:
            out[i] = buf[off: off + int(lens[i]) * hop]
            off += int(lens[i]) * hop
    return out
