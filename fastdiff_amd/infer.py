"""Test-time batching / collation / sharding glue: what `tasks/run.py --infer` does around the hot path.

Reference behaviour restated (none of this is on the GPU path, so plain torch/numpy):
  * `VocoderDataset.load_mel_inputs` (tasks/vocoder/dataset_utils.py:186-204): every `*.npy` under `test_input_dir`, sorted,
    holds a mel of shape [T, 80]; item_name = path below the directory with "/" replaced by "_".
  * `VocoderDataset.collater` at test time (dataset_utils.py:100-160, `batch_max_frames = 0`): the random crop degenerates to
    frames [0, T-1) -- the LAST FRAME IS DROPPED (:116-125, interval_end = 1 so start_frame = 0); mels are zero-padded to the
    longest item of the batch (`collate_2d`, utils/__init__.py:136-150) and transposed to [B, 80, T'].
  * `DistributedSampler(shuffle=False)` (tasks/vocoder/vocoder_base.py:43-49): the index list is padded by wrap-around to a
    multiple of the world size and rank r takes r::world; every rank writes its own wavs, no collective.
  * `test_step` (modules/FastDiff/task/FastDiff.py:96-118): one `sampling_given_noise_schedule` call per batch, then per
    item `wav / wav.abs().max()` and `save_wav` = *32767 -> int16 (utils/audio.py:11-16) as `<item_name>_pred.wav`.
    The reference CLI runs B = 1 (config `max_sentences`); batching is this path's extension: the batch goes down with its
    `lens`, so every item is computed exactly as if it ran alone (and the padding costs nothing), then cropped to its length.

Entry points: load_mel_inputs, load_wav_inputs (device mel front-end), collate_test_batch, distributed_sampler_indices, synthesize, save_wavs and a small CLI
(`python -m fastdiff_amd.infer --test_input_dir D --out_dir O [--N 4] [--ckpt model.ckpt]`).
"""
import argparse
import glob
import os
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import schedules, shard
from .sampler import InferenceSchedule


def load_mel_inputs(test_input_dir: str) -> List[dict]:
    """[{item_name, mel: FloatTensor [T,80], len}] for every *.npy below the directory, in sorted path order."""
    items = []
    for path in sorted(glob.glob(f"{test_input_dir}/*.npy")):
        mel = torch.FloatTensor(np.load(path))
        if mel.dim() != 2:
            raise ValueError(f"{path}: expected a [T, n_mels] array, got shape {tuple(mel.shape)}")
        # the reference keeps the ".npy" suffix in the name (dataset_utils.py:200), so outputs are "<file>.npy_pred.wav"
        items.append({"item_name": path[len(test_input_dir) + 1:].replace("/", "_"), "mel": mel, "len": mel.shape[0]})
    return items


def pcm_to_float(pcm: np.ndarray, what: str = "wav") -> np.ndarray:
    """Samples of a RIFF file as float32 in [-1, 1), the way librosa.core.load (soundfile) hands them to process_utterance
    (data_gen/tts/data_gen_utils.py:100): every integer width is divided by its full scale (uint8 is offset binary, 24-bit PCM
    arrives from scipy left-aligned in int32), float files pass through.  Anything else is refused rather than guessed."""
    if pcm.dtype == np.uint8:
        return (pcm.astype(np.float32) - 128.0) / 128.0
    if pcm.dtype in (np.int16, np.int32):
        return pcm.astype(np.float32) / float(-int(np.iinfo(pcm.dtype).min))
    if pcm.dtype in (np.float32, np.float64):
        wav = pcm.astype(np.float32)
        if wav.size and float(np.abs(wav).max()) > 1.0 + 1e-6:
            raise ValueError(f"{what}: float samples outside [-1, 1] (peak {float(np.abs(wav).max()):.3g})")
        return wav
    raise ValueError(f"{what}: unsupported sample type {pcm.dtype}")


def load_wav_inputs(model, test_input_dir: str, sample_rate: int = 22050, mel_variant: str = "pwg") -> List[dict]:
    """Copy-synthesis inputs (`test_input_dir` with recordings, tasks/vocoder/dataset_utils.py:162-184): every *.wav below the
    directory, in sorted order, through the device mel front-end (`FastDiff.mel_spectrogram` = process_utterance of
    data_gen/tts/data_gen_utils.py:93-147, or with mel_variant="tacotron" the TacotronSTFT of vocoder_binarizer_tacotron.py:110-116
    for models trained on FastDiff_tacotron.yaml features).  Integer PCM is scaled by its full range as librosa.core.load does
    (pcm_to_float); the sample rate must already be the model's (no resampler here)."""
    from scipy.io import wavfile
    items = []
    for path in sorted(glob.glob(f"{test_input_dir}/*.wav")):
        sr, pcm = wavfile.read(path)
        if sr != sample_rate:
            raise ValueError(f"{path}: sample rate {sr}, expected {sample_rate}")
        if pcm.ndim != 1:
            raise ValueError(f"{path}: expected mono audio, got shape {pcm.shape}")
        wav = pcm_to_float(pcm, path)
        mel = model.mel_spectrogram(torch.from_numpy(wav).cuda(), variant=mel_variant)[0].transpose(0, 1).contiguous().cpu()      # [T, 80] as on disk
        items.append({"item_name": path[len(test_input_dir) + 1:].replace("/", "_"), "mel": mel, "len": mel.shape[0]})
    return items


def collate_test_batch(items: Sequence[dict], drop_last_frame: bool = True, out: "np.ndarray" = None):
    """mels [B, 80, T'] zero-padded, lens [B] (frames kept per item), item_names -- the test-time collater.
    The padding and the [T, 80] -> [80, T] transposition are plain numpy slice copies (torch's CPU kernels wake a whole OpenMP team
    for these few hundred KB, which on a 256-core host now and then costs tens of milliseconds).  `out`: optional flat float32
    buffer (e.g. pinned memory) that receives the batch; the returned tensor is then a view of it."""
    kept, lens, names = [], [], []
    for it in items:
        c = it["mel"]
        t = c.shape[0] - 1 if drop_last_frame else c.shape[0]
        if drop_last_frame and c.shape[0] < 2:
            continue                          # the reference cannot collate a one-frame item (dataset_utils.py:110 squeezes it away)
        kept.append(c)
        lens.append(t)
        names.append(it["item_name"])
    if not kept:
        return None, [], []
    B, C, T = len(kept), kept[0].shape[1], max(lens)
    buf = np.empty(B * C * T, np.float32) if out is None else out[: B * C * T]
    batch = buf.reshape(B, C, T)
    for b, (c, t) in enumerate(zip(kept, lens)):
        a = c.numpy() if isinstance(c, torch.Tensor) else np.asarray(c)
        batch[b, :, :t] = a[:t].T
        batch[b, :, t:] = 0.0
    return torch.from_numpy(batch), lens, names


def distributed_sampler_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Indices torch's DistributedSampler(shuffle=False, drop_last=False) hands to `rank`."""
    if n_items == 0:
        return []
    total = -(-n_items // world_size) * world_size
    idx = list(range(n_items))
    while len(idx) < total:                   # padded by repeating from the start (possibly several times)
        idx += idx[: total - len(idx)]
    return idx[rank:total:world_size]


def synthesize(model, items: Sequence[dict], n_steps: int = 4, max_batch: int = 8, seed: int = 0, drop_last_frame: bool = True,
               noise_schedule=None, diffusion_hyperparams=None) -> Dict[str, np.ndarray]:
    """item_name -> int16 PCM of its own length (hop 256 x frames), through length-sorted padded micro-batches.
    Noise: utterance `it` draws x_T and z from Philox stream (seed, it["uid"]) over its own samples (fd_set_noise_streams); "uid"
    defaults to the item's position in `items`, callers that shard a job put the utterance's index in the WHOLE job there, so a
    waveform does not depend on the micro-batch, rank or world size that produced it."""
    if diffusion_hyperparams is None:
        diffusion_hyperparams = schedules.training_hyperparams()
    if noise_schedule is None:
        noise_schedule = schedules.noise_schedule_for(n_steps)
    # the step table depends on the schedule only: derived once (sampling_given_noise_schedule derives it on every call, as the
    # reference does -- 6 ms of host arithmetic per call for N = 6), the same rows then drive every micro-batch
    rows = InferenceSchedule(diffusion_hyperparams, noise_schedule, verbose=False).rows()
    lengths = [it["len"] for it in items]
    out: Dict[str, np.ndarray] = {}
    if not items:
        return out
    # pinned staging, allocated once per job (a pinned allocation per micro-batch costs more than vocoding it): two mel and two PCM
    # buffers, used alternately -- buffer k & 1 is free again once micro-batch k - 2 has been collected, which happens in iteration k - 1
    hop = model.hop_length
    t_max, b_max = max(lengths), min(max_batch, len(items))
    mel_pin = [torch.empty(b_max * 80 * t_max, dtype=torch.float32).pin_memory() for _ in range(2)]
    pcm_pin = [torch.empty(b_max * t_max * hop, dtype=torch.int16).pin_memory() for _ in range(2)]
    mel_np = [m.numpy() for m in mel_pin]      # the collater writes the batch straight into the pinned buffer
    pending = None                 # (event, pinned PCM view, names, lens) of the micro-batch still on its way to the host

    def collect(p):
        done, host, names, lens = p
        done.synchronize()
        for b, (name, t) in enumerate(zip(names, lens)):
            out[name] = host[b, : t * hop].numpy().copy()

    k = 0
    for batch_idx in shard.micro_batches(range(len(items)), lengths, max_batch):
        mels, lens, names = collate_test_batch([items[i] for i in batch_idx], drop_last_frame, out=mel_np[k & 1])
        if mels is None:
            continue
        uid_of = {items[i]["item_name"]: int(items[i].get("uid", i)) for i in batch_idx}
        B, _, T = mels.shape
        mels = mel_pin[k & 1][: B * 80 * T].view(B, 80, T).cuda(non_blocking=True)
        with torch.no_grad():
            wav = model.sample(mels, rows, ddim=False, seed=seed, lens=lens, stream_ids=[uid_of[n] for n in names])
        # one epilogue call and one asynchronous copy per micro-batch; the previous batch is unpacked on the host while this one runs
        pcm = model.peak_normalize_int16(wav, valid=[t * hop for t in lens])
        host = pcm_pin[k & 1][: B * T * hop].view(B, T * hop)
        host.copy_(pcm, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        if pending is not None:
            collect(pending)
        pending = (done, host, names, lens)
        k += 1
    if pending is not None:
        collect(pending)
    return out


def synthesize_sharded(model, items, n_steps: int = 4, max_batch: int = 8, seed: int = 0, drop_last_frame: bool = True, src: int = 0,
                       device=None) -> Dict[str, np.ndarray]:
    """BASELINE config 4 as north_star words it: rank `src` holds all utterances (items; None elsewhere) -> length-balanced
    partition (shard.partition_utterances) -> scatter of the mels -> every rank vocodes its share in padded micro-batches on its
    own GPU -> gather of the int16 PCM on `src`, which returns item_name -> PCM (the other ranks return {}).  The process group
    must be initialised; `device` is where the messages are staged (the GPU for RCCL, None = host for gloo).  Without a process
    group (one process) this is synthesize()."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return synthesize(model, items, n_steps, max_batch, seed, drop_last_frame)
    rank, world = dist.get_rank(), dist.get_world_size()
    meta = [None]
    if rank == src:      # the collater's view of every item: [80, T'] with the last frame dropped, too-short items left out
        kept = [(i, it) for i, it in enumerate(items) if not drop_last_frame or it["mel"].shape[0] >= 2]
        mels = [torch.from_numpy(np.ascontiguousarray(it["mel"].numpy()[: it["mel"].shape[0] - (1 if drop_last_frame else 0)].T)) for _, it in kept]
        meta = [([it["item_name"] for _, it in kept], [int(it.get("uid", i)) for i, it in kept])]
    dist.broadcast_object_list(meta, src=src)
    names, uids = meta[0]
    lens_src = [m.shape[-1] for m in mels] if rank == src else [0] * len(names)
    lens_t = torch.tensor(lens_src, dtype=torch.int64, device=device)
    dist.broadcast(lens_t, src=src)
    parts = shard.partition_utterances(lens_t.tolist(), world)
    mine, lens = shard.scatter_utterances(mels if rank == src else None, parts, src=src, device=device)
    local = [{"item_name": str(i), "mel": m.transpose(0, 1), "len": m.shape[-1], "uid": uids[i]} for i, m in mine]
    pcm = synthesize(model, local, n_steps, max_batch, seed, drop_last_frame=False)
    wavs = [(i, torch.from_numpy(pcm[str(i)]).to(device) if device is not None else torch.from_numpy(pcm[str(i)])) for i, _ in mine]
    out = shard.gather_waveforms(wavs, lens, parts, hop=model.hop_length, dst=src, device=device, dtype=torch.int16)
    if rank != src:
        return {}
    return {names[i]: out[i].cpu().numpy() for i in range(len(names))}


def save_wavs(pcm: Dict[str, np.ndarray], out_dir: str, sample_rate: int = 22050) -> List[str]:
    from scipy.io import wavfile
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for name, data in pcm.items():
        path = os.path.join(out_dir, f"{name}_pred.wav")
        wavfile.write(path, sample_rate, data)
        paths.append(path)
    return paths


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--test_input_dir", required=True, help="directory of [T,80] .npy mels, or of .wav recordings with --from_wav")
    ap.add_argument("--from_wav", action="store_true", help="inputs are recordings: compute the mels on the device first")
    ap.add_argument("--mel_variant", default="pwg", choices=("pwg", "tacotron"), help="front-end for --from_wav (base.yaml / FastDiff_tacotron.yaml features)")
    ap.add_argument("--out_dir", required=True)
    ap.add_argument("--N", type=int, default=4, help="reverse steps: 3, 4, 6, 8, 200 or 1000 (FastDiff.py:76-93)")
    ap.add_argument("--ckpt", default=None, help="reference checkpoint (state_dict under ['state_dict']['model'])")
    ap.add_argument("--max_batch", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args(argv)
    from . import FastDiff
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    torch.manual_seed(args.seed)
    model = FastDiff().cuda().eval()
    if args.ckpt:
        model.load_state_dict(torch.load(args.ckpt, map_location="cpu")["state_dict"]["model"], strict=True)
    items = load_wav_inputs(model, args.test_input_dir, mel_variant=args.mel_variant) if args.from_wav else load_mel_inputs(args.test_input_dir)
    mine = [dict(items[i], uid=i) for i in sorted(set(distributed_sampler_indices(len(items), rank, world)))]
    paths = save_wavs(synthesize(model, mine, args.N, args.max_batch, args.seed), args.out_dir)
    print(f"rank {rank}/{world}: wrote {len(paths)} files to {args.out_dir}")


if __name__ == "__main__":
    main()
