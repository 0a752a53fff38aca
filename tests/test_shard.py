"""Utterance sharding: partition properties, and the scatter -> per-rank work -> gather path on a
2-process gloo group (the CPU stand-in for RCCL; the per-rank "vocoder" here is a deterministic stub)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastdiff_amd import shard


def test_partition_covers_everything_and_balances():
    rng = np.random.default_rng(1234)
    lens = rng.integers(200, 865, size=64).tolist()          # BASELINE config 4: T_i ~ U{200..864}, B=64
    parts = shard.partition_utterances(lens, 8)
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(64))
    loads = [sum(lens[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(lens)               # LPT bound
    assert max(loads) <= 1.08 * (sum(lens) / 8)
    assert shard.partition_utterances(lens, 8) == parts       # deterministic


def test_partition_by_time_counts_the_per_utterance_constant():
    """balance = "time" (shard.utterance_cost): an utterance costs its frames plus a constant, so a rank is not handed twice as many
    short utterances as another gets long ones just because the frames add up."""
    lens = [800] * 4 + [100] * 32                                   # 3200 + 3200 frames
    by_frames = shard.partition_utterances(lens, 2)
    by_time = shard.partition_utterances(lens, 2, cost="time")
    assert sorted(i for p in by_time for i in p) == list(range(36))
    cost = lambda p: sum(shard.utterance_cost(lens[i]) for i in p)      # noqa: E731
    assert abs(cost(by_time[0]) - cost(by_time[1])) <= shard.utterance_cost(800)
    assert abs(cost(by_time[0]) - cost(by_time[1])) <= abs(cost(by_frames[0]) - cost(by_frames[1]))
    # a preloaded rank (the source's own packing / gathering work) is handed less
    pre = shard.partition_utterances([100] * 16, 2, cost="time", preload=[4 * shard.utterance_cost(100), 0.0])
    assert len(pre[0]) == 6 and len(pre[1]) == 10
    assert shard.partition_utterances(lens, 2, cost=lambda t: 1.0) == [sorted(range(0, 36, 2)), sorted(range(1, 36, 2))]


def test_partition_edge_cases():
    assert shard.partition_utterances([], 4) == [[], [], [], []]
    assert shard.partition_utterances([5], 2) == [[0], []]
    assert shard.round_robin_partition(5, 2) == [[0, 2, 4], [1, 3]]   # DistributedSampler(shuffle=False)
    mb = shard.micro_batches([0, 1, 2, 3, 4], [10, 50, 30, 20, 40], 2)
    assert mb == [[1, 4], [2, 3], [0]]


def test_pad_mels_is_collate_2d():
    a, b = torch.ones(80, 3), 2 * torch.ones(80, 5)
    out = shard.pad_mels([a, b])
    assert out.shape == (2, 80, 5) and out[0, :, 3:].abs().sum() == 0 and torch.equal(out[1], b)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _stub_vocoder(mel):
    """[80,T] -> [T*256]: deterministic stand-in for the per-rank HIP vocoder."""
    return mel.mean(0).repeat_interleave(256) + mel.shape[-1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        lens = [5, 9, 3, 7, 4]
        mels = [torch.randn(80, t) for t in lens] if rank == 0 else None
        parts = shard.partition_utterances(lens, world)
        mine, got_lens = shard.scatter_utterances(mels, parts, src=0)
        assert got_lens == lens
        assert sorted(i for i, _ in mine) == parts[rank]
        wavs = [(i, _stub_vocoder(m)) for i, m in mine]
        out = shard.gather_waveforms(wavs, lens, parts, hop=256, dst=0)
        # the on-disk layout [T, 80] travels as it is (no host transposition); lengths already known to every rank
        rows = [m.t().contiguous() for m in mels] if rank == 0 else None
        mine2, _ = shard.scatter_utterances(rows, parts, src=0, lens=lens, frames_first=True)
        assert [i for i, _ in mine2] == [i for i, _ in mine] and all(m.shape == (lens[i], 80) for i, m in mine2)
        assert all(torch.equal(a.t(), b) for (_, a), (_, b) in zip(mine2, mine))
        if rank == 0:
            for i, t in enumerate(lens):
                assert out[i].shape == (t * 256,)
                assert torch.equal(out[i], _stub_vocoder(mels[i]))
            msgs = shard.pack_messages(rows, parts)            # one flat buffer, one slice per rank, utterances back to back
            assert [m.numel() for m in msgs] == [80 * sum(lens[i] for i in p) for p in parts]
            assert torch.equal(msgs[1][: 80 * lens[parts[1][0]]].view(lens[parts[1][0]], 80), rows[parts[1][0]])
            ret.put("ok")
    finally:
        dist.destroy_process_group()


def test_scatter_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get() == "ok"
