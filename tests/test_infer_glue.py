"""Test-time collation / sharding glue: CPU checks against torch's DistributedSampler and a golden batch
produced by the reference's own collate_2d (oracle/gen_golden.py: gen_collate); one GPU run through the whole driver."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from fastdiff_amd import infer, shard


def test_distributed_sampler_indices_match_torch():
    from torch.utils.data.distributed import DistributedSampler
    for n in (1, 2, 3, 7, 8, 9, 17, 64):
        data = list(range(n))
        for world in (1, 2, 3, 4, 8):
            seen = []
            for rank in range(world):
                ref = list(DistributedSampler(data, num_replicas=world, rank=rank, shuffle=False))
                got = infer.distributed_sampler_indices(n, rank, world)
                assert got == ref, (n, world, rank)
                seen += got
            assert set(seen) == set(range(n))                       # every utterance is synthesised by someone
    assert infer.distributed_sampler_indices(0, 0, 4) == []


def test_collate_matches_the_references_collate_2d():
    g = load_golden("collate")
    lens = g["lens"].tolist()
    items = [{"item_name": f"u{i}.npy", "mel": torch.from_numpy(g[f"mel{i}"]), "len": t} for i, t in enumerate(lens)]
    mels, kept, names = infer.collate_test_batch(items, drop_last_frame=False)
    assert torch.equal(mels, torch.from_numpy(g["batch"])) and kept == lens and names == [it["item_name"] for it in items]
    # test time: the collater drops the last frame of every item (dataset_utils.py:116-125) and skips one-frame items
    mels, kept, names = infer.collate_test_batch(items, drop_last_frame=True)
    assert kept == [4, 8, 2, 8] and names == ["u0.npy", "u1.npy", "u2.npy", "u3.npy"]
    assert mels.shape == (4, 80, 8)
    for b, t in enumerate(kept):
        assert torch.equal(mels[b, :, :t], torch.from_numpy(g[f"mel{b}"])[:t].T)
        assert not mels[b, :, t:].any()


def test_directory_to_batch_matches_the_references_own_collater(tmp_path):
    """collater.npz was produced by EXECUTING the reference's VocoderDataset.load_mel_inputs -> __getitem__ -> collater
    (tasks/vocoder/dataset_utils.py:186-204, 80-98, 100-160) with VocoderBinarizer.process_mel_item and collate_2d (oracle/
    gen_golden.py gen_collater) on a directory of [T, 80] mels: same files here -> same item names, sizes and batch tensor."""
    g = load_golden("collater")
    for name in g["file_names"].tolist():
        path = tmp_path / name
        os.makedirs(path.parent, exist_ok=True)
        np.save(path, g["in_" + name.replace("/", "__")])
    items = infer.load_mel_inputs(str(tmp_path))
    assert [it["item_name"] for it in items] == g["item_names"].tolist()          # top level only, sorted, ".npy" kept
    assert [it["len"] for it in items] == g["sizes"].tolist()
    mels, lens, names = infer.collate_test_batch(items)                            # test time: batch_max_frames = 0
    assert names == g["item_names"].tolist()
    assert lens == [t - 1 for t in g["in_lens"].tolist()]                          # the dropped last frame (dataset_utils.py:116-125)
    assert mels.dtype == torch.float32 and torch.equal(mels, torch.from_numpy(g["mels"]))


def test_load_mel_inputs_order_names_and_shapes(tmp_path):
    rng = np.random.default_rng(0)
    for name, t in (("b_second", 7), ("a_first", 5), ("c.third", 3)):
        np.save(tmp_path / f"{name}.npy", rng.standard_normal((t, 80)).astype(np.float32))
    np.save(tmp_path / "ignored.txt.npy.bak", np.zeros(3))                # not *.npy
    os.rename(tmp_path / "ignored.txt.npy.bak.npy", tmp_path / "ignored.bak")
    items = infer.load_mel_inputs(str(tmp_path))
    assert [it["item_name"] for it in items] == ["a_first.npy", "b_second.npy", "c.third.npy"]      # sorted; suffix kept as in the reference
    assert [it["len"] for it in items] == [5, 7, 3] and all(it["mel"].shape[1] == 80 for it in items)
    np.save(tmp_path / "bad.npy", np.zeros(4, np.float32))
    with pytest.raises(ValueError, match="expected a"):
        infer.load_mel_inputs(str(tmp_path))


def test_micro_batches_cover_every_item_once():
    lens = [5, 9, 3, 9, 1, 4, 4]
    batches = shard.micro_batches(range(len(lens)), lens, 3)
    assert sorted(i for b in batches for i in b) == list(range(len(lens))) and all(len(b) <= 3 for b in batches)
    assert [lens[i] for i in batches[0]] == [9, 9, 5]                      # longest first: padding waste stays small


@pytest.mark.gpu
def test_synthesize_directory_end_to_end(tmp_path):
    """Three Tacotron-range mels on disk -> padded micro-batches (with their lens, so every item is computed as if alone:
    tests/test_gpu_parity.py pins that) -> int16 wavs named like the reference's outputs."""
    import fastdiff_amd
    from fastdiff_amd import schedules
    from fastdiff_amd.sampler import sampling_given_noise_schedule
    from scipy.io import wavfile
    rng = np.random.default_rng(3)
    src = tmp_path / "mels"
    src.mkdir()
    for name, t in (("x", 6), ("y", 10), ("z", 8)):
        np.save(src / f"{name}.npy", (rng.random((t, 80)) * 13.5 - 11.5).astype(np.float32))
    torch.manual_seed(1234)
    model = fastdiff_amd.FastDiff().cuda().eval()
    items = infer.load_mel_inputs(str(src))
    pcm = infer.synthesize(model, items, n_steps=4, max_batch=2, seed=11)
    assert {k: v.shape for k, v in pcm.items()} == {"x.npy": (5 * 256,), "y.npy": (9 * 256,), "z.npy": (7 * 256,)}
    assert all(v.dtype == np.int16 and np.abs(v).max() == 32767 for v in pcm.values())
    # the first micro-batch is (y, z): redo it by hand
    mels, lens, names = infer.collate_test_batch([items[1], items[2]])
    assert names == ["y.npy", "z.npy"]          # items 1 and 2 of the sorted directory: their noise streams
    wav = sampling_given_noise_schedule(model, (2, 1, mels.shape[-1] * 256), schedules.training_hyperparams(),
                                        schedules.noise_schedule_for(4), condition=mels.cuda(), seed=11, verbose=False, lens=lens,
                                        stream_ids=[1, 2])
    for b, (name, t) in enumerate(zip(names, lens)):
        own = wav[b, 0, : t * 256]
        ref = (own / own.abs().max() * 32767).cpu().numpy().astype(np.int16)
        assert np.array_equal(pcm[name], ref), name
    # mels already on the GPU (what the RCCL scatter of synthesize_sharded delivers) are collated there; return_device keeps the PCM
    # in HBM (what its gather sends): same bits either way, and a second job on the model reuses the cached table and staging
    dev_items = [dict(it, mel=it["mel"].cuda()) for it in items]
    on_dev = infer.synthesize(model, dev_items, n_steps=4, max_batch=2, seed=11, return_device=True)
    assert all(v.is_cuda and v.dtype == torch.int16 for v in on_dev.values())
    assert all(np.array_equal(on_dev[k].cpu().numpy(), pcm[k]) for k in pcm)
    mixed = infer.synthesize(model, dev_items, n_steps=4, max_batch=3, seed=11)
    assert all(np.array_equal(mixed[k], pcm[k]) for k in pcm)
    assert len(model._infer_cache["rows"]) == 1 and model._infer_cache["pin"] is not None
    paths = infer.save_wavs(pcm, str(tmp_path / "out"))
    assert sorted(os.path.basename(p) for p in paths) == ["x.npy_pred.wav", "y.npy_pred.wav", "z.npy_pred.wav"]
    sr, data = wavfile.read(paths[0])
    assert sr == 22050 and np.array_equal(data, pcm[os.path.basename(paths[0])[:-9]])


@pytest.mark.gpu
def test_copy_synthesis_from_a_recording(tmp_path):
    """wav -> device mel front-end -> vocoder -> wav, through the directory driver (the reference's test_input_dir path)."""
    import fastdiff_amd
    from scipy.io import wavfile
    g = load_golden("frontend_lj001_0002")
    src = tmp_path / "wavs"
    src.mkdir()
    wavfile.write(src / "LJ001-0002.wav", 22050, g["pcm"])
    torch.manual_seed(1234)
    model = fastdiff_amd.FastDiff().cuda().eval()
    items = infer.load_wav_inputs(model, str(src))
    assert [it["item_name"] for it in items] == ["LJ001-0002.wav"] and items[0]["mel"].shape == (164, 80)
    assert np.abs(items[0]["mel"].numpy().T - g["mel_f64"])[g["mel_f64"] > -4.0].max() < 2e-4
    pcm = infer.synthesize(model, items, n_steps=4, max_batch=4, seed=3)
    assert pcm["LJ001-0002.wav"].shape == (163 * 256,) and pcm["LJ001-0002.wav"].dtype == np.int16      # last frame dropped by the collater


@pytest.mark.gpu
def test_synthesize_runs_the_epilogue_again_for_a_call_that_left_the_fp16_range():
    """infer.synthesize drives the pipelined host check (library option fallback = "host", the module's default): a micro-batch whose
    sample call is flagged is run again on the fp32 kernels one batch later, and its int16 epilogue and copy with it.  Forced with a
    first conv scaled by 3e4 (every DBlock / ConvTranspose / LVC launch leaves the fp16 range): the PCM must be what the in-graph
    fallbacks give (option "graph"), host arrays and device tensors alike."""
    import fastdiff_amd
    import synth
    sd = dict(synth.synth_state_dict(1234))
    sd["first_audio_conv.weight_g"] = (sd["first_audio_conv.weight_g"] * 3.0e4).astype(np.float32)
    rng = np.random.default_rng(5)
    items = [{"item_name": f"u{i}", "mel": torch.from_numpy((rng.random((t, 80)) * 7.5 - 6.0).astype(np.float32)), "len": t}
             for i, t in enumerate((9, 14, 6, 11, 7))]
    out = {}
    for mode in ("graph", "host"):
        m = fastdiff_amd.FastDiff()
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        m = m.cuda().eval()
        m.set_option("fallback", mode)
        out[mode] = infer.synthesize(m, items, n_steps=4, max_batch=2, seed=3)
        if mode == "host":
            dev = infer.synthesize(m, [dict(it, mel=it["mel"].cuda()) for it in items], n_steps=4, max_batch=2, seed=3, return_device=True)
    for name, pcm in out["graph"].items():
        assert pcm.dtype == np.int16 and np.abs(pcm).max() == 32767
        # the second pass runs the flagged stages on fp32 in every step, the in-graph form per step: a last-bit difference of the
        # waveform can move a sample by one LSB
        assert np.abs(out["host"][name].astype(np.int32) - pcm.astype(np.int32)).max() <= 1, name
        assert np.array_equal(dev[name].cpu().numpy(), out["host"][name]), name
