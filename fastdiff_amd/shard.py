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


# What one utterance costs a rank, in frames: its own T_i plus a per-utterance constant -- the part of a reverse step that does not
# shrink with the utterance (launch chains of the predictor front, the DBlocks and the hop-8 layers, the epilogue, collation, its
# slice of the PCM copy).  Fitted on one MI355X from fd_sample calls of B = 1..16 at 200..864 frames, N = 6 (tools/cost_model.py,
# profiles/r06_cost_model.json): ms(call) = 1.12 + 0.055 * B + 0.001586 * sum(T_i) (rms residual 0.19 ms), c1 / c2 = 35 frames.  The
# constant PER CALL (c0: 708 frames' worth -- the latency chains of a reverse step that do not shrink with the batch) is what a strong-
# scaling job pays once per rank; the partition cannot balance it away.
UTTERANCE_OVERHEAD_FRAMES = 35.0


def utterance_cost(frames: int, overhead: float = UTTERANCE_OVERHEAD_FRAMES) -> float:
    return float(frames) + overhead


def partition_utterances(lengths: Sequence[int], world_size: int, cost=None, preload: Sequence[float] = None) -> List[List[int]]:
    """Greedy LPT: utterance indices per rank with ~equal total cost.  Deterministic; every index appears once.
    cost: None = frames (T_i); "time" = utterance_cost (frames + the per-utterance constant: a rank that gets many short utterances
    is as busy as one that gets few long ones); or a callable frames -> cost.  preload[r]: cost rank r carries before the first
    utterance is dealt (e.g. the source rank's packing / gathering work, in the same unit), so it is handed less."""
    fn = (lambda t: float(t)) if cost is None else (utterance_cost if cost == "time" else cost)
    costs = [fn(int(t)) for t in lengths]
    order = sorted(range(len(lengths)), key=lambda i: (-costs[i], i))
    loads = [float(v) for v in preload] if preload is not None else [0.0] * world_size
    assert len(loads) == world_size
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += costs[i]
    return [sorted(p) for p in parts]


def round_robin_partition(n_items: int, world_size: int) -> List[List[int]]:
    """The reference's own assignment: DistributedSampler(shuffle=False) gives rank r the indices r::world."""
    return [list(range(r, n_items, world_size)) for r in range(world_size)]


def micro_batches(indices: Sequence[int], lengths: Sequence[int], max_batch: int, sort: bool = True) -> List[List[int]]:
    """Bucket a rank's utterances (longest first) into padded micro-batches of at most `max_batch`.
    sort=False keeps the order of arrival (a stream of requests: the reference CLI's own order, FastDiff.py:97-103)."""
    order = sorted(indices, key=lambda i: (-int(lengths[i]), i)) if sort else list(indices)
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
    if dist.get_backend(ops[0].group) == "gloo":
        # gloo has no pair from a rank to itself (RCCL does): messages to self are matched in posting order and copied
        me = dist.get_rank()
        mine = [op for op in ops if op.peer == me]
        if mine:
            sends, recvs = [op for op in mine if op.op is dist.isend], [op for op in mine if op.op is dist.irecv]
            assert len(sends) == len(recvs)
            for a, b in zip(sends, recvs):
                b.tensor.copy_(a.tensor)
            ops = [op for op in ops if op.peer != me]
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


_pinned = {}


def _pinned_buffer(key: str, n: int, dtype: torch.dtype) -> torch.Tensor:
    """A pinned host buffer of at least n elements, kept between jobs (a pinned allocation costs milliseconds)."""
    buf = _pinned.get(key)
    if buf is None or buf.numel() < n or buf.dtype != dtype:
        buf = torch.empty(n, dtype=dtype).pin_memory()
        _pinned[key] = buf
    return buf


def pack_messages(mels: Sequence[torch.Tensor], parts: List[List[int]], device=None) -> List[torch.Tensor]:
    """All utterances (2-D float32 host tensors, any layout: each travels as its own flat bytes) back to back in the order of
    parts[0], parts[1], ...: ONE buffer, one flat slice per rank.  With `device` a GPU the buffer is pinned host memory, filled by
    numpy slice copies (torch's CPU kernels wake an OpenMP team per call: tens of milliseconds on a 256-core host) and uploaded
    with ONE asynchronous copy; the slices are then device memory, ready for RCCL."""
    import numpy as np
    order = [i for p in parts for i in p]
    sizes = [int(mels[i].numel()) for i in order]
    total = sum(sizes)
    on_gpu = device is not None and torch.device(device).type == "cuda"
    if any(m.is_cuda for m in mels):            # utterances that already live on a GPU (e.g. fresh from the device mel front-end)
        buf = torch.cat([mels[i].reshape(-1).float() for i in order]).to(device if on_gpu else "cpu")
        out, off = [], 0
        for p in parts:
            n = sum(int(mels[i].numel()) for i in p)
            out.append(buf[off: off + n])
            off += n
        return out
    host = _pinned_buffer("scatter", total, torch.float32)[:total] if on_gpu else torch.empty(total, dtype=torch.float32)
    host_np = host.numpy()
    off = 0
    for i, n in zip(order, sizes):
        host_np[off: off + n] = np.asarray(mels[i].detach().numpy(), np.float32).reshape(-1)
        off += n
    buf = host.to(device, non_blocking=True) if on_gpu else host
    out, off = [], 0
    for p in parts:
        n = sum(int(mels[i].numel()) for i in p)
        out.append(buf[off: off + n])
        off += n
    return out


def scatter_utterances(mels: Sequence[torch.Tensor], parts: List[List[int]], src: int = 0, device=None, lens: Sequence[int] = None,
                       frames_first: bool = False, loopback: bool = False):
    """Rank `src` holds all mels; afterwards every rank holds its own (index, mel) list.
    Lengths travel first as one small broadcast (skipped when the caller already distributed them: `lens`); then ONE packed
    message per peer (its utterances back to back, in the order of parts[r]) in a single grouped send/recv -- at most world-1
    messages leave `src`, whatever the number of utterances.  Layout: [80, T_i] tensors by default; frames_first = the on-disk
    layout [T_i, 80] (dataset_utils.py:186-204), which saves the host transposition -- the receiver's collater transposes on the
    device.  Returns (own (index, mel) list, all lengths).
    loopback (a world of ONE rank, `parts` of any length): parts[1:] are "virtual peers" -- their packed messages go through the
    backend's grouped send/recv from this rank to itself and are unpacked from the receive buffers, so the transport and the
    packing / unpacking of a multi-rank job run on a box with a single GPU; the rank ends up with every utterance."""
    rank, world = dist.get_rank(), dist.get_world_size()
    assert not loopback or world == 1, "loopback is the single-rank stand-in for peers"
    n = sum(len(p) for p in parts)
    t_axis = 0 if frames_first else -1
    if lens is None:
        lens_t = torch.zeros(n, dtype=torch.int64, device=device)
        if rank == src:
            lens_t = torch.tensor([m.shape[t_axis] for m in mels], dtype=torch.int64, device=device)
        dist.broadcast(lens_t, src=src)
        lens = lens_t.tolist()
    lens_l = [int(v) for v in lens]

    def unpack(buf, idx):
        mine, off = [], 0
        for i in idx:
            m = buf[off: off + 80 * lens_l[i]]
            mine.append((i, m.view(lens_l[i], 80) if frames_first else m.view(80, lens_l[i])))
            off += 80 * lens_l[i]
        return mine

    if rank == src:
        msgs = pack_messages(mels, parts, device)
        if loopback:
            back = {r: torch.empty_like(msgs[r]) for r in range(1, len(parts)) if parts[r]}
            _exchange([op for r, b in back.items() for op in (dist.P2POp(dist.isend, msgs[r], src), dist.P2POp(dist.irecv, b, src))])
            return unpack(msgs[0], parts[0]) + [m for r, b in back.items() for m in unpack(b, parts[r])], lens_l
        _exchange([dist.P2POp(dist.isend, msgs[r], r) for r in range(world) if r != src and parts[r]])
        return unpack(msgs[src], parts[src]), lens_l
    if not parts[rank]:
        return [], lens_l
    buf = torch.empty(80 * sum(lens_l[i] for i in parts[rank]), dtype=torch.float32, device=device)
    _exchange([dist.P2POp(dist.irecv, buf, src)])
    return unpack(buf, parts[rank]), lens_l


def gather_waveforms(mine, lens: Sequence[int], parts: List[List[int]], hop: int = 256, dst: int = 0, device=None,
                     dtype: torch.dtype = torch.float32, loopback: bool = False):
    """Inverse of scatter_utterances: rank `dst` ends up with the list of waveforms ([T_i*hop] each, float32 or the int16 PCM of
    the device epilogue) in index order.  One packed message per peer, all receives of `dst` posted as one group.
    loopback: see scatter_utterances -- the waveforms of parts[1:] come back as packed byte messages from this rank to itself."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if loopback:
        assert world == 1 and rank == dst
        by_idx = dict(mine)
        out = [None] * len(lens)
        for i in parts[0]:
            out[i] = by_idx[i]
        sends = {r: torch.cat([by_idx[i].reshape(-1).to(dtype) for i in parts[r]]) for r in range(1, len(parts)) if parts[r]}
        if device is not None:
            sends = {r: b.to(device) for r, b in sends.items()}
        recvs = {r: torch.empty_like(b) for r, b in sends.items()}
        _exchange([op for r in sends for op in (dist.P2POp(dist.isend, _as_bytes(sends[r]), dst), dist.P2POp(dist.irecv, _as_bytes(recvs[r]), dst))])
        for r, buf in recvs.items():
            off = 0
            for i in parts[r]:
                out[i] = buf[off: off + int(lens[i]) * hop]
                off += int(lens[i]) * hop
        return out
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
