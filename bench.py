#!/usr/bin/env python
"""bench.py -- FastDiff vocoder inference on MI355X: real-time factor of the N=4 reverse sampler.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: fd_sample() of B=8 utterances of 80x864 mel (10.03 s each)
through N=4 reverse steps (BASELINE.json configs[1]), mel resident in HBM, waveform left in HBM.  With N GPUs every
rank runs its own batch (independent utterances, no data-path collective): weak scaling, value = whole-job audio
seconds per wall second.  Prints ONE JSON line on rank 0.

Extra objects in the line:
  roofline     -- the dominant kernel of the step, timed live with HIP events on the launch stream (library option
                  "profile"), algorithmic bytes/flops per launch from DESIGN.md;
  cpu_baseline -- the CPU oracle (a C port of the reference algorithm, oracle/) timed on this host on a bounded
                  sample: one utterance (B=1, T=864), N=4.  The reference's own PyTorch CPU path cannot be timed on the
                  GPU box (/root/reference is absent there); its timing in the build container is in DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR, HOP = 22050, 256
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix = fp32 vector peak
MFMA_F16_PEAK_TFLOPS = 2500.0 # dense fp16/bf16 matrix peak (MI355X_MICROARCH.md)


def kernel_model(name, B, T):
    """Algorithmic work of ONE launch of a kernel family: (bound, bytes, flops)."""
    L = T * HOP
    if name.startswith("lvc_layer_h"):
        hop = int(name.split("_h")[1].split("_")[0])
        # read x, skip (32 ch each), write x (32 ch) at rate hop*T; read the frame's 64x96 kernel + 64 biases
        return "hbm", 4.0 * B * T * (96 * hop + 6208), 2.0 * B * T * hop * (32 * 96 + 64 * 96)
    if name == "kp_gemm":
        # all 3 LVC blocks in one launch: [24832 x 192] x [192 x B*T] each, output written once (fp32 matrix pipe)
        return "mfma", 3 * 4.0 * (B * T * 24832 + 24832 * 192 + B * T * 64), 3 * 2.0 * 24832 * 192 * B * T
    if name == "kp_gemm_f16x2":
        # the same product as three fp16 MFMA passes (2-piece operands): 3x the flops on a 16x faster pipe -- the
        # 2.06 GB of predicted kernels it writes is what bounds it
        return "hbm", 3 * 4.0 * (B * T * 24832 + 24832 * 192 + B * T * 64), 3 * 3 * 2.0 * 24832 * 192 * B * T
    if name == "kp_front":
        # input conv (K=400) + six 64->64 k3 convs (K=192) for the three predictors, fused through LDS
        return "mfma", 3 * 4.0 * B * T * (80 + 64), 3 * 2.0 * 64 * (400 + 6 * 192) * B * T
    if name.startswith("dblock"):
        return "hbm", None, None
    if name.startswith("convt"):
        return "hbm", None, None
    if name == "first_conv":
        return "hbm", 4.0 * B * L * 33, 2.0 * 7 * 32 * B * L
    if name == "final_conv_update":
        return "hbm", 4.0 * B * L * 34, 2.0 * 7 * 32 * B * L
    if name == "final_update":      # the conv itself ran inside the last LVC layer: read + clear the sums, read + write x
        return "hbm", 4.0 * B * L * 4, 0.0
    return "hbm", None, None


def family(name):
    if os.environ.get("FD_BENCH_SPLIT"):
        return name
    for p in ("lvc_layer_h8", "lvc_layer_h64", "lvc_layer_h256"):
        if name.startswith(p + "_"):
            return p
    return name


def measure_roofline(model, mel, rows, B, T, nsteps, lens=None):
    """Eager (graph off) profiled pass: per-kernel HIP-event timing on the launch stream."""
    model.set_option("profile", "1")
    try:
        with torch.no_grad():
            model.sample(mel, rows, seed=1, lens=lens)
            torch.cuda.synchronize()
            model.profile(reset=True)
            for _ in range(2):
                model.sample(mel, rows, seed=1, lens=lens)
            torch.cuda.synchronize()
        stats = model.profile(reset=True)
    finally:
        model.set_option("profile", "0")
    fam = {}
    for name, (launches, ms) in stats.items():
        f = fam.setdefault(family(name), [0, 0.0])
        f[0] += launches
        f[1] += ms
    total_ms = sum(v[1] for v in fam.values())
    table = {}
    for name, (launches, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        bound, nbytes, flops = kernel_model(name, B, T)
        avg_ms = ms / launches
        e = {"launches": launches, "avg_us": round(avg_ms * 1e3, 2), "share": round(ms / total_ms, 4)}
        if nbytes:
            e["GBps"] = round(nbytes / (avg_ms * 1e-3) / 1e9, 1)
        if flops:
            e["TFLOPs"] = round(flops / (avg_ms * 1e-3) / 1e12, 2)
        table[name] = e
    dom = next(iter(table))
    bound, nbytes, flops = kernel_model(dom, B, T)
    avg_s = fam[dom][1] / fam[dom][0] * 1e-3
    if bound == "mfma" and flops:
        roof = {"kernel": dom, "bound": "mfma", "achieved": round(flops / avg_s / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s"}
    else:
        roof = {"kernel": dom, "bound": "hbm", "achieved": round((nbytes or 0.0) / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    roof["avg_launch_us"] = round(avg_s * 1e6, 2)
    roof["traffic"] = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # filled from rocprofv3 --pmc passes, see profiles/README.md
    if os.path.exists(pmc):
        try:
            roof["traffic"] = json.load(open(pmc)).get(dom)
        except Exception:
            pass
    return roof, table


def cpu_baseline(T, rows):
    """One utterance, N=len(rows) steps, on the host cores, two ways: the network restated with PyTorch CPU ops
    (oracle/torch_eager.py -- the ATen conv1d / conv_transpose1d / einsum calls the reference itself makes, pinned on its goldens),
    which is the reported baseline, and the plain-C OpenMP port of the oracle beside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import synth
    from oracle import Oracle
    from torch_eager import EagerFastDiff
    N = len(rows)
    cores = os.cpu_count() or 1
    audio_s = T * HOP / SR
    # --- PyTorch CPU eager
    threads_t = min(cores, 32)
    prev = torch.get_num_threads()
    torch.set_num_threads(threads_t)
    try:
        m = EagerFastDiff(synth.synth_state_dict(1234))
        mel_t = torch.from_numpy(synth.synth_mel(1, 1, T))
        x_t = torch.from_numpy(synth.hash_normal(1, 1, T * HOP).reshape(1, 1, T * HOP))
        with torch.no_grad():
            m.sample(mel_t[:, :, :32], rows, x_t[:, :, : 32 * HOP])          # warm the thread pool and the op caches
            t0 = time.perf_counter()
            m.sample(mel_t, rows, x_t)
            dt_t = time.perf_counter() - t0
    finally:
        torch.set_num_threads(prev)
    # --- C port
    o = Oracle("f32")
    # parallelism of the port is over (batch, output channel) = 32..64 rows: more threads than that only add contention
    threads = o.set_threads(min(cores, 32))
    o.set_weights(synth.synth_state_dict(1234))
    mel = synth.synth_mel(1, 1, T)
    x_T = synth.hash_normal(1, 1, T * HOP).reshape(1, 1, T * HOP)
    z = np.zeros((N, 1, 1, T * HOP), np.float32)
    ex = rows[::-1]   # oracle tables are indexed by reverse index n
    table = {"steps": [r["t"] for r in ex], "c_eps": [r["c_eps"] for r in ex], "c_div": [r["c_div"] for r in ex],
             "sigma_hat": [r["sigma"] for r in ex], "c1": [r["c1"] for r in ex], "c2": [r["c2"] for r in ex],
             "c3": [r["c3"] for r in ex]}
    o.forward(synth.synth_audio(1, 1, 16), synth.synth_mel(1, 1, 16), np.zeros(1, np.float32))   # warm the thread pool
    t0 = time.perf_counter()
    o.sample(mel, table, x_T, z)
    dt = time.perf_counter() - t0
    return {"value": round(audio_s / dt_t, 3), "unit": "x real-time", "cores": threads_t, "kind": "port",
            "sample": f"oracle/torch_eager.py (torch {torch.__version__} CPU ops, fp32, {threads_t} threads of {cores} cores) B=1 T={T} N={N}: {dt_t:.2f} s wall",
            "samples_per_s": round(T * HOP / dt_t, 1),
            "c_port": {"value": round(audio_s / dt, 3), "cores": threads,
                       "sample": f"oracle/fastdiff_oracle.c (fp32, OpenMP {threads} threads) B=1 T={T} N={N}: {dt:.2f} s wall"}}


def torch_eager_baseline(mel, rows, audio_s, reps=3):
    """The same N-step sampling as plain PyTorch-ROCm eager ops on the same GPU (oracle/torch_eager.py: conv1d / conv_transpose1d /
    unfold + einsum through MIOpen and rocBLAS, fp32, weights from the same seed): what running the reference's PyTorch code on
    this box amounts to.  A reported baseline like cpu_baseline, never the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import synth
    from torch_eager import EagerFastDiff
    m = EagerFastDiff(synth.synth_state_dict(1234), device=mel.device)
    B, _, T = mel.shape
    with torch.no_grad():
        x_T = torch.randn(B, 1, T * HOP, device=mel.device)
        m.sample(mel, rows, x_T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = m.sample(mel, rows, x_T)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
    assert torch.isfinite(out).all()
    return {"ms_per_step": round(ms, 3), "value": round(audio_s / (ms / 1e3), 2), "unit": "x real-time", "kind": "port",
            "sample": "oracle/torch_eager.py, torch %s eager fp32 on this GPU, B=%d T=%d N=%d, %d repetitions" % (torch.__version__, B, T, len(rows), reps)}


def host_inclusive(model, mel, rows, lens, audio_s, reps=5):
    """SURVEY.md 8d's wall clock: mel resident on the HOST -> int16 waveform resident on the HOST (pinned buffers, PCIe both ways,
    the waveform epilogue on the device).  Reported beside `value`, never as `value`: the boundary takes device pointers."""
    mel_h = mel.cpu().pin_memory()
    B, _, T = mel.shape
    pcm_h = torch.empty((B, T * HOP), dtype=torch.int16).pin_memory()
    with torch.no_grad():
        def one(i):
            m = mel_h.to(mel.device, non_blocking=True)
            pcm = model.peak_normalize_int16(model.sample(m, rows, seed=i, lens=lens))
            pcm_h.copy_(pcm, non_blocking=True)
        one(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            one(1 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
    return {"ms_per_step": round(ms, 4), "value": round(audio_s / (ms / 1e3), 2), "unit": "x real-time",
            "path": "pinned host mel -> device -> fd_sample -> fd_peak_normalize_int16 -> pinned host int16 PCM"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=864)
    ap.add_argument("--nsteps", type=int, default=4, help="reverse steps N (3,4,6,8,200,1000)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ragged", action="store_true",
                    help="BASELINE config 4 style batch: T_i ~ U{200..frames}, zero-padded; RTF counts the valid audio only")
    ap.add_argument("--no-lens", action="store_true", help="with --ragged: do not tell the library the lengths (padded compute)")
    ap.add_argument("--torch-eager-baseline", action="store_true",
                    help="also time the plain PyTorch-ROCm eager restatement of the same sampling on this GPU (off by default)")
    ap.add_argument("--no-host-io", action="store_true", help="skip the extra host-to-host (PCIe-inclusive) measurement")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="library option (fd_set_option), repeatable")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: fastdiff_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import fastdiff_amd
    from fastdiff_amd import sampler, schedules

    B, T, N = args.batch, args.frames, args.nsteps
    torch.manual_seed(1234)                       # weights = FastDiff() default init, seed 1234
    model = fastdiff_amd.FastDiff().to(dev).eval()
    if args.no_graph:
        model.set_option("graph", "0")
    for kv in args.opt:
        model.set_option(*kv.split("=", 1))
    torch.manual_seed(1234 + rank)
    mel = (torch.rand(B, 80, T) * 7.5 - 6.0).to(dev)    # uniform on [mel_vmin, mel_vmax]
    lens = None
    valid_frames = B * T
    if args.ragged:
        lens = torch.randint(200, T + 1, (B,)).tolist() if T > 200 else [T] * B
        for b, t in enumerate(lens):
            mel[b, :, t:] = 0.0                          # collate_2d padding
        valid_frames = sum(lens)
    dh = schedules.training_hyperparams()
    rows = sampler.InferenceSchedule(dh, schedules.noise_schedule_for(N), verbose=False).rows()

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])

    with torch.no_grad():
        for i in range(args.warmup):
            out = model.sample(mel, rows, seed=i, lens=None if args.no_lens else lens)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = model.sample(mel, rows, seed=100 + i, lens=None if args.no_lens else lens)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(out if lens is None else torch.stack([out[b, :, : lens[b] * HOP].abs().max() for b in range(B)])).all()
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    audio_s = world * valid_frames * HOP / SR      # (ragged: rank 0's draw stands for every rank)
    line = {
        "metric": "real-time factor (audio-sec/wall-sec), N=%d reverse steps, 80x%d mel" % (N, T),
        "value": round(audio_s / (ms_per_step / 1e3), 2),
        "unit": "x real-time",
        "samples_per_s": round(world * B * T * HOP / (ms_per_step / 1e3), 1),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: LJSpeech FastDiff.yaml shape, batch=%d utterances of 80x%d mel, "
                               "N=%d, HIP LVC/dilated-conv kernels + hipGraph sampler" % (B, T, N),
                   "batch_per_gpu": B, "frames": T, "reverse_steps": N, "sharding": "utterances/rank, no collective",
                   "graph": not args.no_graph, "weights": "random init seed 1234 (no checkpoint offline)",
                   "ragged": (None if not args.ragged else {"lens": lens, "told_to_library": not args.no_lens})},
    }
    if rank == 0 and world == 1 and not args.no_host_io:
        line["host_inclusive"] = host_inclusive(model, mel, rows, None if args.no_lens else lens, audio_s)
    if rank == 0 and world == 1:
        if not args.no_roofline:
            roof, table = measure_roofline(model, mel, rows, B, T, N, None if args.no_lens else lens)
            line["roofline"] = roof
            line["kernels"] = table
        if args.torch_eager_baseline:
            try:
                line["torch_eager_baseline"] = torch_eager_baseline(mel, rows, audio_s)
            except Exception as e:
                line["torch_eager_baseline"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(T, rows)
            except Exception as e:   # the checker is optional for the measurement
                line["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
