"""The N > 1 path of BASELINE config 4: rank 0 holds every utterance -> length-balanced partition -> scatter ->
per-rank padded micro-batches -> gather of the int16 PCM.  `infer.synthesize_sharded` on 2-process gloo groups:
  * CPU: the per-rank vocoder replaced by a deterministic stand-in (partition / packing / naming / dtype logic);
  * GPU: the real HIP vocoder in both processes (both on cuda:0, gloo moving the messages) -- every waveform bit-equal to the
    single-process result, which per-utterance noise streams (fd_set_noise_streams) and `lens` make possible.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastdiff_amd import infer


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _items(seed, lens):
    g = torch.Generator().manual_seed(seed)
    return [{"item_name": f"utt{i:02d}.npy", "mel": torch.rand(t, 80, generator=g) * 7.5 - 6.0, "len": t} for i, t in enumerate(lens)]


def _stub_pcm(mel_T80, uid):
    """Deterministic stand-in for fd_sample + the int16 epilogue: depends on the mel AND on the noise-stream id."""
    t = mel_T80.shape[0]
    base = (mel_T80.mean(1) * 1000).to(torch.int16).repeat_interleave(256)
    return (base + torch.tensor(uid, dtype=torch.int16)).numpy()


def _stub_synthesize(model, items, n_steps=4, max_batch=8, seed=0, drop_last_frame=True, **kw):
    out = {}
    for i, it in enumerate(items):
        mel = it["mel"][: it["mel"].shape[0] - 1] if drop_last_frame else it["mel"]
        out[it["item_name"]] = _stub_pcm(mel, int(it.get("uid", i)))
    return out


class _Hop:
    hop_length = 256


def _cpu_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        infer.synthesize = _stub_synthesize
        lens = [9, 4, 13, 2, 7, 7, 1, 5]          # item 6 has one frame: the collater drops it (dataset_utils.py:116-125)
        items = _items(3, lens) if rank == 0 else None
        out = infer.synthesize_sharded(_Hop(), items, n_steps=4, max_batch=3, seed=5, drop_last_frame=True, src=0, device=None)
        if rank == 0:
            kept = [(i, it) for i, it in enumerate(items) if it["len"] >= 2]
            assert sorted(out) == sorted(it["item_name"] for _, it in kept)
            for uid, it in kept:
                want = _stub_pcm(it["mel"][:-1], uid)          # uid = the item's index in the whole job, as in one process
                assert out[it["item_name"]].dtype == np.int16 and np.array_equal(out[it["item_name"]], want), it["item_name"]
            ret.put("ok")
        else:
            assert out == {}
    finally:
        dist.destroy_process_group()


def test_synthesize_sharded_world2_gloo_stub_vocoder():
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get() == "ok"


def _gpu_worker(rank, world, port, ret, gpu_lock):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpu_common
        torch.cuda.set_device(0)                    # both ranks share the one GPU of the test box; gloo carries the messages
        model = gpu_common.make_model()
        lens = [40, 12, 33, 7, 25, 18, 40, 3, 29]
        items = _items(11, lens) if rank == 0 else None
        if os.environ.get("FD_TEST_SERIALIZE", "1") == "1":
            # the two ranks take turns on the one GPU of the test box (see the note at the test below)
            real = infer.synthesize

            def one_rank_at_a_time(*a, **k):
                with gpu_lock:
                    r = real(*a, **k)
                    torch.cuda.synchronize()
                    return r
            infer.synthesize = one_rank_at_a_time
        out = infer.synthesize_sharded(model, items, n_steps=4, max_batch=2, seed=77, drop_last_frame=True, src=0, device=None)
        if rank == 0:
            single = infer.synthesize(model, items, n_steps=4, max_batch=4, seed=77, drop_last_frame=True)
            assert sorted(out) == sorted(single)
            bad = [name for name in single if not np.array_equal(out[name], single[name])]
            if bad:      # seen about once in 30 runs of this two-processes-on-one-GPU arrangement: say which side is the odd one
                again = infer.synthesize(model, items, n_steps=4, max_batch=4, seed=77, drop_last_frame=True)
                dump = os.path.join(root, "gpurun_out")
                os.makedirs(dump, exist_ok=True)
                for name in bad:
                    a, b, c = (v[name].astype(np.int32) for v in (out, single, again))
                    d = np.abs(a - b)
                    idx = np.nonzero(d)[0]
                    scale = float((a * b).sum()) / max(float((b * b).sum()), 1.0)
                    print(f"MISMATCH {name}: {idx.size} of {d.size} samples, max |d| {d.max()}, first {idx[0]}, last {idx[-1]}; "
                          f"least-squares scale sharded/single {scale:.6f}, residual after scaling {np.abs(a - scale * b).max():.1f}; "
                          f"a second single-process run equals: sharded {np.array_equal(a, c)}, single {np.array_equal(b, c)}", flush=True)
                    np.savez_compressed(os.path.join(dump, f"sharded_mismatch_{name}.npz"), sharded=out[name], single=single[name], again=again[name])
            assert not bad, bad
            assert all(v.dtype == np.int16 for v in out.values())
            other = infer.synthesize(model, items, n_steps=4, max_batch=4, seed=78, drop_last_frame=True)
            assert not np.array_equal(other[items[0]["item_name"]], single[items[0]["item_name"]])      # the seed matters
            assert "fastdiff_amd/lib/libfastdiff_hip.so" in open("/proc/self/maps").read()
            ret.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_synthesize_sharded_world2_hip_vocoder_equals_single_process():
    """Both ranks drive the real HIP vocoder on cuda:0 (the test box has one GPU); every waveform of the sharded job must be bit-equal
    to the single-process one.  The ranks take turns on the GPU: with both processes vocoding AT THE SAME TIME on the one device, about
    3 % of runs showed one utterance off in a few hundred samples (the last 64 columns of one 256-column tile of a hop-256 layer read
    stale; also with graph=0 and fuse_final=0; never in one process, never with the ranks taking turns: 0 of 110 runs against 5 of 171 --
    profiles/r02/s12_s19_two_processes_one_gpu.txt).  One process per GPU, which is what the sharded path is for, has no second
    process on its device; FD_TEST_SERIALIZE=0 restores the concurrent arrangement for hunting (tools/gpu_r2_s14.sh)."""
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    gpu_lock = ctx.Lock()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, ret, gpu_lock)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert ret.get() == "ok"


def test_pcm_to_float_scales_every_integer_width():
    assert np.allclose(infer.pcm_to_float(np.array([-32768, 0, 16384], np.int16)), [-1.0, 0.0, 0.5])
    assert np.allclose(infer.pcm_to_float(np.array([-2 ** 31, 2 ** 30], np.int32)), [-1.0, 0.5])
    assert np.allclose(infer.pcm_to_float(np.array([0, 128, 255], np.uint8)), [-1.0, 0.0, 127 / 128])
    assert infer.pcm_to_float(np.array([0.25, -1.0], np.float64)).dtype == np.float32
    with pytest.raises(ValueError, match="outside"):
        infer.pcm_to_float(np.array([1.5], np.float32))
    with pytest.raises(ValueError, match="unsupported"):
        infer.pcm_to_float(np.array([1], np.int64))
