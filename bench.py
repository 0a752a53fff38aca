#!/usr/bin/env python
"""bench.py -- FastDiff vocoder inference on MI355X: real-time factor of the N-step reverse sampler.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload configs1|config4|config5]

`--gpus N` with N > 1 launches the N ranks itself (one process per GPU under torch.distributed.run, the reference's own
one-process-per-GPU model: utils/trainer.py:94-107 mp.spawn) unless it already runs under a launcher (WORLD_SIZE set).  It refuses
to run when the box has fewer than N GPUs: it never prints a line for a world it did not measure.  `n_gpus` in the line is the live
RCCL world size (an all-reduce of ones), not the flag.

Workloads
  configs1 (default) -- BASELINE.json configs[1], the configuration the metric is quoted on: every rank runs fd_sample() on its own
            batch of B=8 utterances of 80x864 mel (10.03 s each), N=4; weak scaling, no data-path collective.  One "step" = pinned host
            mel -> device -> one fd_sample call -> int16 epilogue -> pinned host PCM (the metric is defined host to
            host); the same steps with the mel resident in HBM and the waveform left there: `value_device_resident`.
  config4 -- BASELINE.json configs[3] as north_star words it: rank 0 holds 64 ragged utterances (T_i ~ U{200..864}, N=6) on the
            HOST; one step = length-balanced partition -> scatter of the mels (one packed RCCL message per peer) -> per-rank padded
            micro-batches through fd_sample + the int16 epilogue -> gather of the PCM on rank 0's host.  Total work is fixed:
            strong scaling; the time is host-to-host.

  config5 -- BASELINE.json configs[4]: a directory of 16 Tacotron-range mels stored [T, 80] as .npy (ln(clamp(., 1e-5)) range, T_i in
            300..864) -> the test-time loader and collater (dataset_utils.py:186-204, :100-160) -> one padded batch of 16 through
            fd_sample (N=4) + the int16 epilogue -> PCM on the host.  One step = the whole job from the files to host int16.

Extra objects in the JSON line (rank 0, N=1 only where they need one GPU):
  roofline       -- the dominant kernel of the step, timed live with HIP events on the launch stream (library option "profile"),
                    algorithmic bytes/flops per launch from DESIGN.md section 3;
  host_inclusive -- SURVEY.md 8(d)'s wall clock for the same batch: pinned host mel -> device -> fd_sample -> int16 epilogue ->
                    pinned host PCM (PCIe both ways);
  fp32_pipe      -- the same step with every contraction on the exact-fp32 matrix instruction (options gemm/lvc/conv = fp32):
                    what the default pipe (2-piece fp16 operands, fp32 accumulation) buys;
  parity         -- max |difference| of both pipes to the float64 CPU oracle on one utterance (B=1, T=864, N=4, injected x_T);
  cpu_baseline   -- the CPU restatement of the reference (kind "port": /root/reference does not exist on the GPU box) timed on
                    this host on a bounded sample: one utterance (B=1, T=864), N=4.
"""
import argparse
import json
import os
import socket
import subprocess
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
DTYPE = "f32 results; 2-piece fp16 operands (22 bit) on fp16 MFMA, fp32 accumulate; GEMM as Winograd F(2,3); exact-fp32 twin"


def kernel_model(name, B, T):
    """Algorithmic work of ONE launch of a kernel family: (bound, bytes, flops).  `bytes` is what THIS launch has to
    move at least (SURVEY.md 8(d)'s figure for the plain LVC layer; a fused launch is charged only what it really needs)."""
    L = T * HOP
    if name.startswith("lvc_up_h"):
        # the first layer of blocks 1 / 2 with the block's ConvTranspose inside: reads the block input (32 ch at rate hop/r, r = 8 / 4)
        # instead of x, skip and the frame record as every layer, writes x
        hop = int(name.split("_h")[1].split("_")[0])
        r = 4 if hop == 256 else 8
        return "hbm", 4.0 * B * T * (64 * hop + 32 * hop // r + 6208), 2.0 * B * T * hop * (32 * 96 + 64 * 96 + 32 * 64)
    if name.startswith("lvc_final_h"):
        # the last layer of the last block with final_conv inside: reads x, skip and the record, writes ONE channel of per-sample sums
        # (eps_acc, 4 B per sample) instead of its 32 output channels
        hop = int(name.split("_h")[1].split("_")[0])
        return "hbm", 4.0 * B * T * (64 * hop + 6208) + 4.0 * B * L, 2.0 * B * T * hop * (32 * 96 + 64 * 96) + 2.0 * 7 * 32 * B * L
    if name.startswith("lvc_layer_h"):
        hop = int(name.split("_h")[1].split("_")[0])
        # read x, skip (32 ch each), write x (32 ch) at rate hop*T; read the frame's 64x96 kernel + 64 biases: SURVEY.md 8(d)
        return "hbm", 4.0 * B * T * (96 * hop + 6208), 2.0 * B * T * hop * (32 * 96 + 64 * 96)
    if name == "kp_gemm":
        # all 3 LVC blocks in one launch: [24832 x 192] x [192 x B*T] each, output written once (fp32 matrix pipe)
        return "mfma", 3 * 4.0 * (B * T * 24832 + 24832 * 192 + B * T * 64), 3 * 2.0 * 24832 * 192 * B * T
    if name == "kp_gemm_f16x2":
        # the same product on the fp16 pipe with 2-piece operands (three MFMA passes: 3x these flops are EXECUTED, see
        # executed_flops); the 2.06 GB of predicted kernels it writes is what bounds it
        return "hbm", 3 * 4.0 * (B * T * 24832 + 24832 * 192 + B * T * 64), 3 * 2.0 * 24832 * 192 * B * T
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


GEMM_FORM = "winograd"      # set from the library option gemm_form in main()


def executed_flops(name, B, T):
    """Matrix-pipe flops a launch really issues when that differs from the algorithmic count: the fp16x2 GEMM forms h.h, h.l and l.h
    (3x), and as Winograd F(2,3) over the frame axis four K = 64 products per pair of frames instead of six (x 2/3)."""
    if name == "kp_gemm_f16x2":
        return 3 * kernel_model(name, B, T)[2] * (2.0 / 3.0 if GEMM_FORM == "winograd" else 1.0)
    return None


def unfused_bytes(name, B, T):
    """What the ops a fused launch replaces would move as separate launches (for the 'credited with the un-fused ops' bytes' figure):
    lvc_up = the ConvTranspose (read 32 ch at hop/r, write 32 ch at hop) + a plain layer; lvc_final = a plain layer."""
    if name.startswith("lvc_up_h"):
        hop = int(name.split("_h")[1].split("_")[0])
        r = 4 if hop == 256 else 8
        return 4.0 * B * T * (96 * hop + 6208) + 4.0 * B * T * (32 * hop // r + 32 * hop)
    if name.startswith("lvc_final_h"):
        hop = int(name.split("_h")[1].split("_")[0])
        return 4.0 * B * T * (96 * hop + 6208)
    return kernel_model(name, B, T)[1]


def family(name):
    if os.environ.get("FD_BENCH_SPLIT"):
        return name
    for p in ("lvc_layer_h8", "lvc_layer_h64", "lvc_layer_h256"):
        if name.startswith(p + "_"):
            return p
    return name


def template_group(fam_name):
    """Rows that are instantiations of one kernel template: k_lvc_h2<HOP, DIL, FINAL, UP> runs as lvc_layer_h<HOP> (plain), lvc_final_h256
    (final_conv inside, no output channels written) and lvc_up_h<HOP> (the block's ConvTranspose inside)."""
    for p in ("lvc_layer_h", "lvc_final_h", "lvc_up_h"):
        if fam_name.startswith(p):
            return "lvc_h" + fam_name[len(p):]
    return fam_name


ROCPROF_NAME = {"k_kp_gemm_h2": "kp_gemm_f16x2", "k_kp_gemm_w": "kp_gemm_f16x2", "k_h_wino": "h_wino", "k_final_acc": "final_update", "k_first_conv": "first_conv", "k_kp_front_h2": "kp_front",
                "k_advance": "advance_step", "k_init_noise": "init_noise", "k_embed_mlp": "embed", "k_embed_fct": "embed_fct", "k_kp_gemm": "kp_gemm",
                "k_final": "final_conv_update", "k_h_split": "h_split"}


def rocprof_row(name):
    """Kernel name of a rocprofv3 trace -> the row it has in this file's `kernels` table (None: not one of this library's kernels)."""
    import re
    m = re.search(r"k_lvc_h2<(\d+), *\d+, *(\w+), *(\d+)", name)      # <HOP, DIL, FINAL, UP>
    if m:
        if int(m.group(3)) > 0:
            return "lvc_up_h" + m.group(1)
        return ("lvc_final_h" if m.group(2) in ("true", "1") else "lvc_layer_h") + m.group(1)
    if "k_lvc_h8m<" in name:
        return "lvc_layer_h8"
    m = re.search(r"k_dblock_h2<(\d+)", name)
    if m:
        return "dblock_f" + m.group(1)
    m = re.search(r"k_convt_h2<(\d+)", name)
    if m:
        return "convt_r" + m.group(1)
    m = re.search(r"fdk\w*::(k_\w+)", name)
    return ROCPROF_NAME.get(m.group(1), m.group(1)) if m else None


def rocprof_graph_replay(args, calls=5):
    """The timed loop of this very command (same batch, frames, schedule, options; graph replays, nothing else) run once more in a
    child process under `rocprofv3 --kernel-trace --stats`: {row: [launches, total ms]} + the number of sample calls, or None when
    rocprofv3 is not there / failed.  These are the durations a committed kernel_stats.csv of the command holds."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="fd_bench_prof_", dir="/tmp")
    warm = 2
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "kt", "--", sys.executable, os.path.abspath(__file__),
           "--steps", str(calls), "--warmup", str(warm), "--batch", str(args.batch), "--frames", str(args.frames), "--nsteps", str(args.nsteps)]
    for kv in args.opt:
        cmd += ["--opt", kv]
    env = dict(os.environ, FD_BENCH_CHILD="1", TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return {"error": "rocprofv3 child rc=%d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        fam = {}
        for row in csv.DictReader(open(files[0])):
            k = rocprof_row(row["Name"])
            if k is None:
                continue
            f = fam.setdefault(k, [0, 0.0])
            f[0] += int(row["Calls"])
            f[1] += float(row["TotalDurationNs"]) * 1e-6
        keep = os.environ.get("FD_BENCH_KEEP_STATS")
        if keep:
            shutil.copy(files[0], keep)
        return {"fam": fam, "calls": warm + calls}
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_roofline(model, mel, rows, B, T, nsteps, lens=None, replay=None):
    """Per-kernel durations of the step, two ways:
    * `replay` (rocprof_graph_replay): the kernels inside the REPLAYED hipGraph of the timed loop, from a rocprofv3 kernel trace of
      a child run -- the step as it is timed, and what a committed rocprofv3 summary of this command shows.  `roofline` and the
      `kernels` table are computed from these when they are there;
    * in this process, graph off: every launch through hipExtLaunchKernelGGL with start / stop events that receive the dispatch's
      own begin / end timestamps (library option profile = 1), launches back to back on the launch stream.  Reported beside the
      first as `avg_us_eager` (3-7 % shorter: each timestamped dispatch completes -- caches written back -- before the next starts).
      FD_BENCH_PROFILE=events selects hipEventRecord around each launch instead."""
    reps = 3
    mode = os.environ.get("FD_BENCH_PROFILE", "1")
    model.set_option("profile", mode)
    try:
        with torch.no_grad():
            model.sample(mel, rows, seed=1, lens=lens)
            torch.cuda.synchronize()
            model.profile(reset=True)
            for _ in range(reps):
                model.sample(mel, rows, seed=1, lens=lens)
            torch.cuda.synchronize()
        stats = model.profile(reset=True)
    finally:
        model.set_option("profile", "0")
    eager = {}
    for name, (launches, ms) in stats.items():
        f = eager.setdefault(family(name), [0, 0.0])
        f[0] += launches
        f[1] += ms
    timing_eager = ("dispatch begin/end timestamps (hipExtLaunchKernelGGL events), launches back to back, graph off" if mode == "1"
                    else "hipEventRecord around each launch, graph off")
    if replay and replay.get("fam"):
        fam, reps = replay["fam"], replay["calls"]
        timing = "rocprofv3 kernel trace of the timed loop re-run in a child process (kernels inside the replayed graph, %d calls)" % reps
    else:
        fam, timing = eager, timing_eager
    total_ms = sum(v[1] for v in fam.values())
    table = {}
    for name, (launches, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        bound, nbytes, flops = kernel_model(name, B, T)
        avg_ms = ms / launches
        e = {"launches": launches, "launches_per_step": round(launches / (reps * nsteps), 3), "avg_us": round(avg_ms * 1e3, 2), "share": round(ms / total_ms, 4)}
        if nbytes:
            e["MB"] = round(nbytes / 1e6, 1)
            e["GBps"] = round(nbytes / (avg_ms * 1e-3) / 1e9, 1)
            e["hbm_frac"] = round(nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if flops:
            e["TFLOPs_algorithmic"] = round(flops / (avg_ms * 1e-3) / 1e12, 2)
            ex = executed_flops(name, B, T)
            if ex:
                e["TFLOPs_executed"] = round(ex / (avg_ms * 1e-3) / 1e12, 2)
        if fam is not eager and name in eager:
            e["avg_us_eager"] = round(eager[name][1] / eager[name][0] * 1e3, 2)
        table[name] = e
    # the dominant kernel = the kernel TEMPLATE with the largest share of the step; its roofline is quoted on the plain instantiation
    groups = {}
    for name, (launches, ms) in fam.items():
        groups[template_group(name)] = groups.get(template_group(name), 0.0) + ms
    dom_group = max(groups, key=groups.get)
    members = [k for k in fam if template_group(k) == dom_group]
    plain = [k for k in members if k.startswith("lvc_layer_h")]
    dom = plain[0] if plain else max(members, key=lambda k: fam[k][1])
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
    roof["algorithmic_MB_per_launch"] = round((nbytes or 0.0) / 1e6, 1)
    roof["timing"] = timing
    if fam is not eager and dom in eager:
        avg_e = eager[dom][1] / eager[dom][0] * 1e-3
        roof["eager"] = {"avg_launch_us": round(avg_e * 1e6, 2), "frac": round((nbytes or 0.0) / avg_e / 1e9 / HBM_PEAK_GBS, 4), "timing": timing_eager}
    if replay and replay.get("error"):
        roof["replay_error"] = replay["error"]
    roof["template_share_of_step"] = round(groups[dom_group] / total_ms, 4)
    if len(members) > 1:
        roof["kernel_is"] = "plain instantiation of the dominant template %s; each instantiation's own byte model under `variants`" % dom_group
        roof["variants"] = {k: {kk: table[k][kk] for kk in ("launches_per_step", "avg_us", "MB", "GBps", "hbm_frac", "share") if kk in table[k]} for k in sorted(members)}
    # HBM bytes per launch come from rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, each its own run: tools/gpu_round.sh), which
    # cannot run inside this process: the committed summary of the same command is quoted, with its source, or null
    roof["traffic"] = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            d = json.load(open(pmc))
            if d.get(dom) is not None:
                roof["traffic"] = d.get(dom)
                roof["traffic_source"] = ("profiles/pmc_traffic.json: %s" % d.get("_source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes"))[:120]
                roof["traffic_measured_here"] = False      # a committed figure of an earlier session, not of this run
                if "variants" in roof:
                    for k in roof["variants"]:
                        if d.get(k) is not None:
                            roof["variants"][k]["traffic"] = d[k]
        except Exception:
            pass
    # the LVC layers time-weighted (north_star's ">= 50 % of HBM roofline in the LVC kernel" spans all twelve launches per step)
    def weighted(keys, byte_fn=lambda k: kernel_model(k, B, T)[1]):
        by = sum(byte_fn(k) * fam[k][0] for k in keys)
        ms = sum(fam[k][1] for k in keys)
        return {"GBps": round(by / (ms * 1e-3) / 1e9, 1), "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "ms_per_sample_call": round(ms / reps, 4), "launches_per_step": sum(fam[k][0] for k in keys) // (reps * nsteps)}
    lvc = [k for k in fam if k.startswith("lvc_layer_h")]
    fused = [k for k in fam if k.startswith("lvc_up_h") or k.startswith("lvc_final_h")]
    if lvc:
        roof["lvc_plain_layers"] = weighted(lvc)
    if lvc and fused:
        allk = lvc + fused
        a = weighted(allk)
        a["frac_minimal_bytes"] = a.pop("frac")
        a["GBps_minimal_bytes"] = a.pop("GBps")
        u = weighted(allk, lambda k: unfused_bytes(k, B, T))
        a["frac_unfused_ops_bytes"] = u["frac"]
        # SURVEY.md 8(d)'s own unit: every launch is one "LVC layer call" = 4 B T (96 hop + 6208) bytes, fused or not
        sv = weighted(allk, lambda k: 4.0 * B * T * (96 * int(k.split("_h")[1].split("_")[0]) + 6208))
        a["frac_survey_8d_bytes"] = sv["frac"]
        a["note"] = ("all LVC launches of a step, time-weighted: plain layers + lvc_final_h256 (final_conv inside) + lvc_up_h64/h256 (the block's "
                     "ConvTranspose inside).  minimal = each launch charged what it must move itself; unfused = the fused launches credited "
                     "with the bytes of the separate ops they replace; survey_8d = SURVEY.md 8(d)'s per-layer-call figure for every launch")
        roof["lvc_all_12_launches"] = a
        roof["lvc12_frac_min_bytes"] = a["frac_minimal_bytes"]           # short scalars: what the driver's record keeps of this object
        roof["lvc12_frac_survey_bytes"] = a["frac_survey_8d_bytes"]
    for k in ("lvc_layer_h8", "lvc_layer_h64", "lvc_up_h64", "lvc_up_h256", "lvc_final_h256"):
        if k in table and "hbm_frac" in table[k]:
            roof["frac_" + k.replace("lvc_", "").replace("layer_", "")] = table[k]["hbm_frac"]      # on the launch's OWN minimal bytes
            if not k.startswith("lvc_layer_h"):      # ... and on SURVEY 8(d)'s unit: the launch is one LVC layer call, 4 B T (96 hop + 6208) bytes
                b8d = 4.0 * B * T * (96 * int(k.split("_h")[1]) + 6208)
                roof["frac8d_" + k.replace("lvc_", "")] = round(b8d / (table[k]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    if "kp_gemm_f16x2" in table:
        roof["gemm_us"] = table["kp_gemm_f16x2"]["avg_us"]
        roof["gemm_hbm_frac"] = table["kp_gemm_f16x2"].get("hbm_frac")
        roof["gemm_tflops_executed"] = table["kp_gemm_f16x2"].get("TFLOPs_executed")
    return roof, table


def _oracle_table(rows):
    ex = rows[::-1]   # oracle tables are indexed by reverse index n
    return {"steps": [r["t"] for r in ex], "c_eps": [r["c_eps"] for r in ex], "c_div": [r["c_div"] for r in ex],
            "sigma_hat": [r["sigma"] for r in ex], "c1": [r["c1"] for r in ex], "c2": [r["c2"] for r in ex],
            "c3": [r["c3"] for r in ex]}


PORT_VS_REFERENCE = "port = 0.89x / 0.71x (8 / 1 threads) of the reference's own time, outputs 4e-6 apart (profiles/r06_ref_vs_port_cpu.json, round 6)"


def cpu_baseline(T, rows):
    """SURVEY.md 8(d)'s recipe on this host: one utterance (B=1, T frames), N=len(rows) steps, 1 warm-up + best of 3, with
    torch.set_num_threads(os.cpu_count()), and the same at 32 threads and at 1 thread.  What is timed is the PORT of the reference
    (oracle/torch_eager.py: the ATen conv1d / conv_transpose1d / einsum calls the reference itself makes, pinned on its goldens) --
    /root/reference does not exist on the GPU box; PORT_VS_REFERENCE says what the substitution is worth.  The plain-C OpenMP port of
    the oracle is timed beside it.  `value` = the best of the three thread counts (the baseline is not handicapped)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import synth
    from oracle import Oracle
    from torch_eager import EagerFastDiff
    N = len(rows)
    cores = os.cpu_count() or 1
    audio_s = T * HOP / SR
    prev = torch.get_num_threads()
    m = EagerFastDiff(synth.synth_state_dict(1234))
    mel_t = torch.from_numpy(synth.synth_mel(1, 1, T))
    x_t = torch.from_numpy(synth.hash_normal(1, 1, T * HOP).reshape(1, 1, T * HOP))
    by_threads = {}
    try:
        skipped = {}
        for th in sorted({cores, min(cores, 32), 1}, reverse=True):
            torch.set_num_threads(th)
            reps = 3 if th > 1 else 2      # (one thread: ~10 s per run; best of 2 keeps the default bench inside its minutes)
            if th > 32:
                # On the pool's 256-core hosts torch's OpenMP team of all cores costs ~150 ms per small op (the run took 223 s, RTF 0.045:
                # profiles/r05/s2_*): probe one small convolution first and leave the leg out -- with the probe's figure -- when it is that
                # slow, so that the default bench stays inside its minutes.  The 32-thread leg is the CPU's best case either way.
                xp, wp = torch.randn(1, 32, 4096), torch.randn(32, 32, 3)
                torch.nn.functional.conv1d(xp, wp, padding=1)
                t0 = time.perf_counter()
                for _ in range(10):
                    torch.nn.functional.conv1d(xp, wp, padding=1)
                per_op = (time.perf_counter() - t0) / 10
                if per_op > 0.01:
                    skipped[th] = "left out: one 32x32x3 conv1d over 4096 samples takes %.0f ms at %d threads (thread-team overhead; ~1500 such ops per run)" % (per_op * 1e3, th)
                    continue
            with torch.no_grad():
                m.sample(mel_t[:, :, :32], rows, x_t[:, :, : 32 * HOP])          # warm-up: thread pool, op caches
                best = float("inf")
                for _ in range(reps):
                    t0 = time.perf_counter()
                    m.sample(mel_t, rows, x_t)
                    best = min(best, time.perf_counter() - t0)
            by_threads[th] = best
    finally:
        torch.set_num_threads(prev)
    th_best = min(by_threads, key=by_threads.get)
    # --- C port
    o = Oracle("f32")
    # parallelism of the port is over (batch, output channel) = 32..64 rows: more threads than that only add contention
    threads = o.set_threads(min(cores, 32))
    o.set_weights(synth.synth_state_dict(1234))
    mel = synth.synth_mel(1, 1, T)
    x_T = synth.hash_normal(1, 1, T * HOP).reshape(1, 1, T * HOP)
    z = np.zeros((N, 1, 1, T * HOP), np.float32)
    table = _oracle_table(rows)
    o.forward(synth.synth_audio(1, 1, 16), synth.synth_mel(1, 1, 16), np.zeros(1, np.float32))   # warm the thread pool
    t0 = time.perf_counter()
    o.sample(mel, table, x_T, z)
    dt = time.perf_counter() - t0
    out = {"value": round(audio_s / by_threads[th_best], 3), "unit": "x real-time", "cores": th_best, "kind": "port",
           "sample": "torch_eager port, B=1 T=%d N=%d, warm-up + best of 3 (2 at 1 thread); best = %d threads of %d cores" % (T, N, th_best, cores),
           "threads_tried": sorted(list(by_threads) + list(skipped)),
           "samples_per_s": round(T * HOP / by_threads[th_best], 1), "port_vs_reference": PORT_VS_REFERENCE}
    for th, dt_t in by_threads.items():
        out["rtf_%dt" % th] = round(audio_s / dt_t, 3)
        out["s_%dt" % th] = round(dt_t, 3)
    for th, why in skipped.items():
        out["rtf_%dt" % th] = None
        out["note_%dt" % th] = why
    out["rtf_c_port_%dt" % threads] = round(audio_s / dt, 3)
    return out


def parity_vs_oracle(T, rows, dev):
    """Both pipes of the product against the float64 CPU oracle on ONE utterance at the benchmark length (B=1, T frames, N steps,
    injected x_T, zero z): the error figure that belongs next to the `dtype` string.  The oracle is the checker, outside any timing."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import synth
    from oracle import Oracle
    import fastdiff_amd
    N = len(rows)
    sd = synth.synth_state_dict(1234)
    o = Oracle("f64")
    o.set_threads(min(os.cpu_count() or 1, 32))
    o.set_weights(sd)
    mel = synth.synth_mel(1, 1, T)
    x_T = synth.hash_normal(1, 1, T * HOP).reshape(1, 1, T * HOP)
    ref = o.sample(mel, _oracle_table(rows), x_T, np.zeros((N, 1, 1, T * HOP), np.float32))
    out = {"sample": f"B=1 T={T} N={N}, synthetic weights (oracle/synth.py seed 1234), injected x_T, z = 0; float64 oracle; max|x_0| = {float(np.abs(ref).max()):.3f}"}
    m = fastdiff_amd.FastDiff()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.to(dev).eval()
    zeros = torch.zeros((N, 1, 1, T * HOP), device=dev)
    for label, opts in (("f16x2", {"gemm": "f16x2", "lvc": "f16x2", "conv": "f16x2"}), ("fp32", {"gemm": "fp32", "lvc": "fp32", "conv": "fp32"})):
        for k, v in opts.items():
            m.set_option(k, v)
        with torch.no_grad():
            y = m.sample(torch.from_numpy(mel).to(dev), rows, x_T=torch.from_numpy(x_T).to(dev), noise=zeros)
        torch.cuda.synchronize()
        out["max_abs_diff_" + label] = float(np.abs(y.cpu().numpy().astype(np.float64) - ref).max())
    del m
    return out


def fp32_pipe(model, mel, rows, lens, audio_s, reps=5):
    """The same sample call with every contraction on v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain)."""
    for k in ("gemm", "lvc", "conv"):
        model.set_option(k, "fp32")
    try:
        with torch.no_grad():
            model.sample(mel, rows, seed=0, lens=lens)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                model.sample(mel, rows, seed=1 + i, lens=lens)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
    finally:
        for k in ("gemm", "lvc", "conv"):
            model.set_option(k, "f16x2")
    return {"ms_per_step": round(ms, 4), "value": round(audio_s / (ms / 1e3), 2), "unit": "x real-time",
            "options": "gemm=fp32 lvc=fp32 conv=fp32"}


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
    res = {"ms_per_step": round(ms, 3), "value": round(audio_s / (ms / 1e3), 2), "unit": "x real-time", "kind": "port",
           "sample": "oracle/torch_eager.py, torch %s eager fp32 on this GPU, B=%d T=%d N=%d, %d repetitions" % (torch.__version__, B, T, len(rows), reps)}
    # ... and with what the reference's own code adds around the same kernels on every call: weight-norm evaluated inside every
    # convolution, the schedule recursion and the 1000-iteration step-mapping loop per level on the host, CPU random numbers copied
    # to the device per step (still a restatement: /root/reference cannot travel to this box)
    from fastdiff_amd import sampler, schedules
    m2 = EagerFastDiff(synth.synth_state_dict(1234), device=mel.device, weight_norm_each_forward=True)
    dh = schedules.training_hyperparams()
    sched = schedules.noise_schedule_for(len(rows))
    with torch.no_grad():
        m2.sample_like_the_reference(mel, dh, sched, sampler._map_noise_scale_to_time_step_loop)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out2 = m2.sample_like_the_reference(mel, dh, sched, sampler._map_noise_scale_to_time_step_loop)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) / reps * 1e3
    assert torch.isfinite(out2).all()
    res["like_the_reference"] = {"ms_per_step": round(ms2, 3), "value": round(audio_s / (ms2 / 1e3), 2), "kind": "port",
                                 "adds": "weight-norm in every conv call, per-call schedule recursion + step-mapping loops on the host, CPU std_normal + H2D per step"}
    return res


def host_inclusive(model, mel, rows, lens, audio_s, reps=20):
    """SURVEY.md 8d's wall clock: mel resident on the HOST -> int16 waveform resident on the HOST (pinned buffers, PCIe both ways,
    the waveform epilogue on the device).  Reported beside `value`, never as `value`: the boundary takes device pointers."""
    mel_h = mel.cpu().pin_memory()
    B, _, T = mel.shape
    pcm_h = torch.empty((B, T * HOP), dtype=torch.int16).pin_memory()
    with torch.no_grad():
        def one(i):
            m = mel_h.to(mel.device, non_blocking=True)
            wav = model.sample(m, rows, seed=i, lens=lens, defer_check=True)
            pcm_h.copy_(model.peak_normalize_int16(wav), non_blocking=True)
            return model.last_ticket, wav

        def settle(prev):      # option fallback = host: a call that had to be redone on fp32 gets its epilogue and copy again
            if prev is not None and model.settle(prev[0]):
                pcm_h.copy_(model.peak_normalize_int16(prev[1]), non_blocking=True)
        one(0)
        model.check()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prev = None
        for i in range(reps):
            cur = one(1 + i)
            settle(prev)
            prev = cur
        settle(prev)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        # the two PCIe legs on their own (boxes of the pool differ here by an order of magnitude)
        pcm_d = torch.empty((B, T * HOP), dtype=torch.int16, device=mel.device)
        legs = {}
        for name, fn, nbytes in (("h2d_mel", lambda: mel_h.to(mel.device, non_blocking=True), mel_h.numel() * 4),
                                 ("d2h_pcm", lambda: pcm_h.copy_(pcm_d, non_blocking=True), pcm_h.numel() * 2)):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            legs[name] = {"ms": round(dt * 1e3, 4), "GBps": round(nbytes / dt / 1e9, 2)}
    return {"ms_per_step": round(ms, 4), "value": round(audio_s / (ms / 1e3), 2), "unit": "x real-time",
            "path": "pinned host mel -> device -> fd_sample -> fd_peak_normalize_int16 -> pinned host int16 PCM", "pcie": legs}


def b1_object(model, mel, rows, steps):
    """The reference CLI's own mode (max_sentences 1: base.yaml:53, FastDiff.py:101-103): ONE utterance per sample call, same
    length and schedule, mel resident in HBM, plus the dominant kernel's roofline fraction at that size."""
    m1 = mel[:1].contiguous()
    T = m1.shape[-1]
    with torch.no_grad():
        for i in range(3):
            model.sample(m1, rows, seed=i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            model.sample(m1, rows, seed=50 + i, defer_check=True)
        model.check()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        # what the host spends inside one sample() call: enqueue only -- the device is idle when the call starts and the range check
        # is settled outside the stamped region (in the pipelined loop above a call also waits for its predecessor's flags)
        host = []
        for i in range(max(steps, 10)):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            model.sample(m1, rows, seed=90 + i, defer_check=True)
            host.append(time.perf_counter() - h0)
            model.check()
        host.sort()
    roof, table = measure_roofline(model, m1, rows, 1, T, len(rows))
    top = {k: {"avg_us": v["avg_us"], "share": v["share"], **({"hbm_frac": v["hbm_frac"]} if "hbm_frac" in v else {})} for k, v in list(table.items())[:8]}
    return {"ms_per_step": round(ms, 4), "value": round(T * HOP / SR / (ms / 1e3), 2), "unit": "x real-time", "batch": 1, "frames": T,
            "host_us_per_call_median": round(host[len(host) // 2] * 1e6, 1),      # time the host spends inside sample() (enqueue only)
            "roofline": {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "timing") if k in roof},
            "lvc_all_12_launches_frac_minimal_bytes": roof.get("lvc_all_12_launches", {}).get("frac_minimal_bytes"), "kernels_top": top}


def box_state():
    """Clocks / power of this rank's GPU as rocm-smi reports them (the pool's boxes differ by +-7 %: a number without them
    cannot be compared with another session's)."""
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        card = d.get("card0", next(iter(d.values())))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "power" in kl or "performance level" in kl:
                keep[k] = v
        return keep
    except Exception as e:      # noqa: BLE001 -- diagnostics only
        return {"error": repr(e)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_spawn(n):
    """Re-execute this command under torch.distributed.run with one rank per GPU and return its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def config4_items(seed=1234, n=64, t_lo=200, t_hi=864):
    """BASELINE configs[3] / SURVEY.md 8(d): 64 utterances, T_i ~ U{200..864} (seeded), mels uniform on [mel_vmin, mel_vmax], stored
    [T, 80] as on disk (tasks/vocoder/dataset_utils.py:186-204)."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(t_lo, t_hi + 1, (n,), generator=g).tolist()
    return [{"item_name": "utt%03d" % i, "mel": torch.rand(t, 80, generator=g) * 7.5 - 6.0, "len": t} for i, t in enumerate(lens)]


def run_config4(args, model, rank, world, local_rank, dev):
    """One step = the whole sharded job, host to host (see the module docstring)."""
    from fastdiff_amd import infer
    items = config4_items() if rank == 0 else None
    N = args.nsteps
    oversub = dist.is_initialized() and dist.get_backend() == "gloo"
    stage_dev = dev if (world > 1 and not oversub) else None      # RCCL moves device buffers, gloo host buffers

    def one(i):
        return infer.synthesize_sharded(model, items, n_steps=N, max_batch=args.batch, seed=1234 + i, drop_last_frame=False, src=0, device=stage_dev,
                                        gather=args.gather, balance=args.balance)

    def barrier():
        if world > 1:
            dist.barrier() if oversub else dist.barrier(device_ids=[local_rank])

    for i in range(args.warmup):
        out = one(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one(100 + i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    frames = 0
    if rank == 0:
        by_name = {it["item_name"]: it for it in items}
        if args.gather == "src" or world == 1:
            assert sorted(out) == sorted(by_name)
        for name, a in out.items():      # (gather = none: this rank's own share)
            assert a.shape == (by_name[name]["len"] * HOP,) and int(abs(a).max()) == 32767
        frames = sum(it["len"] for it in items)
    if world > 1 and args.gather == "none":      # every utterance on exactly one rank
        n_mine = torch.tensor([len(out)], dtype=torch.int64, device=None if oversub else dev)
        dist.all_reduce(n_mine)
        assert int(n_mine.item()) == 64, int(n_mine.item())
    extra = None
    if world == 1 and args.project_ranks > 1:
        extra = project_sharded(args, model, items, elapsed / args.steps, dev)
    if world > 1:
        # the sharded job against the single-process job on this rank's GPU: per-utterance noise streams make every waveform
        # independent of the rank, micro-batch and world size that produced it -- all of them must be bit-equal
        same = None
        if rank == 0:
            ref = infer.synthesize(model, items, N, args.batch, 1234 + 100 + args.steps - 1, drop_last_frame=False)
            import numpy as np
            same = (args.gather == "none" or sorted(ref) == sorted(out)) and all(np.array_equal(ref[k], out[k]) for k in out)
            extra = {"waveforms_on_rank0": len(out), "bit_equal_to_single_process_job": bool(same), "partition": "LPT (%s) into %d parts" % (args.balance, world),
                     "messages": "%d packed mel messages out, %s" % (world - 1, "none back (every rank keeps its share)" if args.gather == "none"
                                                                      else "%d packed PCM messages back" % (world - 1))}
            assert same, "sharded job differs from the single-process job"
        barrier()
    return elapsed, frames, extra


def config5_files(tmp, seed=50, n=16):
    """BASELINE configs[4] / SURVEY.md 8(d)(5): 16 Tacotron-range mels -- uniform on the range of ln(clamp(., 1e-5)),
    tacotron/audio_processing.py:78-84 -- written [T, 80] float32 .npy as Tacotron leaves them (dataset_utils.py:186-204 reads them
    back); the lengths of tests/test_gpu_parity.py::test_config5_tacotron_batch16_through_the_driver (300..864 frames)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lens = rng.integers(300, 865, n).tolist()
    lens[3], lens[11] = 300, 864
    for i, t in enumerate(lens):
        np.save(os.path.join(tmp, "taco%02d.npy" % i), (rng.random((t, 80)) * 13.5 - 11.5).astype(np.float32))
    return lens


def run_config5(args, model):
    """One step = the whole job: the .npy files of a directory -> load_mel_inputs -> the test-time collater (drops the last frame,
    dataset_utils.py:116-125) -> one padded batch of 16 with `lens` through fd_sample (N steps) -> device int16 epilogue -> PCM on the
    host.  Reported beside it: the same with the 16 wav files written (utils/audio.py:11-16 save_wav)."""
    import shutil
    import tempfile
    from fastdiff_amd import infer
    tmp = tempfile.mkdtemp(prefix="fd_config5_", dir="/tmp")
    try:
        lens = config5_files(tmp)

        def one(i, save=False):
            items = infer.load_mel_inputs(tmp)
            pcm = infer.synthesize(model, items, n_steps=args.nsteps, max_batch=args.batch, seed=1234 + i)
            if save:
                infer.save_wavs(pcm, os.path.join(tmp, "wav"))
            return pcm
        for i in range(args.warmup):
            out = one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = one(100 + i)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        one(0, save=True)
        t1 = time.perf_counter()
        for i in range(args.steps):
            one(200 + i, save=True)
        torch.cuda.synchronize()
        with_wav = (time.perf_counter() - t1) / args.steps
        assert len(out) == len(lens)
        for name, a in out.items():
            t = lens[int(name[4:6])] - 1
            assert a.shape == (t * HOP,) and a.dtype.name == "int16" and int(abs(a).max()) == 32767, name
        frames = sum(t - 1 for t in lens)
        extra = {"utterances": len(lens), "frames_min_max": [min(lens) - 1, max(lens) - 1], "padded_frames": len(lens) * (max(lens) - 1),
                 "ms_per_job_with_16_wav_files_written": round(with_wav * 1e3, 3),
                 "parity": "tests/test_gpu_parity.py::test_config5_tacotron_batch16_through_the_driver: 3 items within 1 int16 LSB of the f64 oracle"}
        return elapsed, frames, extra
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_contractive_(model, lam=1.0, gain=0.5):
    """Random-init weights give a denoiser whose output has nothing to do with its input: over a 1000-step schedule |x| grows without
    bound (2e7 with seed 1234) and leaves the fp16 range half-way -- no trained model does that.  This edits final_conv IN PLACE so
    that eps is positively correlated with x, as a trained denoiser's is: the network's one linear path from x to eps is
    4 * final_conv * first_audio_conv (the skip a0 is added to the residual stream in each of the last block's four layers,
    modules.py:208-209); adding lam times the time-reversed folded first-conv taps to final_conv's direction makes that cascade's centre
    tap positive, and weight_g = gain sets its norm.  |x| then contracts to <= ~1.3 within 125 steps and ends inside [-1, 1]
    (the same construction as the parity fixture's weights: oracle/synth.py make_contractive, tests/golden/sample_s7.npz)."""
    with torch.no_grad():
        sd = model.state_dict()
        w1v, w1g = sd["first_audio_conv.weight_v"], sd["first_audio_conv.weight_g"]
        w1 = w1g * w1v / w1v.pow(2).sum(dim=(1, 2), keepdim=True).sqrt()
        sd["final_conv.0.weight_v"] = sd["final_conv.0.weight_v"] + lam * w1[:, 0, :].flip(-1)[None]
        sd["final_conv.0.weight_g"] = torch.full_like(sd["final_conv.0.weight_g"], gain)
        model.load_state_dict(sd)
    return model


def stream_items(seed=4321, n=256, t_lo=200, t_hi=864):
    """The reference CLI's own call pattern (modules/FastDiff/task/FastDiff.py:97-103 with config/base.yaml:53 max_valid_sentences 1,
    tasks/vocoder/dataset_utils.py:114-125): one utterance per sampling call and a DIFFERENT length each time.  n utterances whose
    lengths are all distinct, drawn without replacement from {t_lo..t_hi} (seeded), in the order of arrival."""
    g = torch.Generator().manual_seed(seed)
    lens = (torch.randperm(t_hi - t_lo + 1, generator=g)[:n] + t_lo).tolist()
    return [{"item_name": "req%03d" % i, "mel": torch.rand(t, 80, generator=g) * 7.5 - 6.0, "len": t} for i, t in enumerate(lens)]


def run_stream(args, model, quick=False):
    """One step = one pass over the whole stream of 256 requests, host mel -> host int16 PCM (infer.synthesize: pinned staging, the
    int16 epilogue on the device, the previous request collected while the next one runs).  Legs:
      b1 / b1_again   one request per fd_sample call in the order of arrival, first pass and a second pass over the same stream (what
                      the graph cache has kept by then is all the difference between the two);
      b1_no_graph     the same with option graph = 0 (every kernel launched by itself: nothing to capture, nothing to miss);
      b8              length-sorted micro-batches of 8 (padded, `lens`): 32 calls;
      b1_sync         the reference's loop body with nothing pipelined: upload, sample, normalise, download, wait -- per request;
      *_exact_T_graphs  option t_bucket = 0: one graph per exact T, i.e. one capture per request (what round 5 shipped);
      fixed           the control: 256 calls of ONE length (the stream's mean), i.e. every call replays a warm graph.
    quick (the object the DEFAULT bench line carries): the control, b1 (two passes), b8 and b1_sync only, N = 4."""
    from fastdiff_amd import infer
    items = stream_items(n=args.stream_requests)
    N = 4 if quick else args.nsteps
    lens = [it["len"] for it in items]
    frames = sum(lens)
    audio_s = frames * HOP / SR
    t_mean = int(round(frames / len(lens)))
    fixed = [{"item_name": "fix%03d" % i, "mel": items[i]["mel"].new_empty(t_mean, 80).uniform_(-6.0, 1.5), "len": t_mean} for i in range(len(items))]

    def one_pass(its, batch, seed, sort):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = infer.synthesize(model, its, n_steps=N, max_batch=batch, seed=seed, drop_last_frame=False, sort=sort)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert len(out) == len(its)
        for it in its[:: max(1, len(its) // 8)]:
            a = out[it["item_name"]]
            assert a.shape == (it["len"] * HOP,) and int(abs(a).max()) == 32767, it["item_name"]
        return dt

    def leg(its, batch, sort, passes, audio):
        ts = [one_pass(its, batch, 7 + k, sort) for k in range(passes)]
        return {"ms_per_request": [round(t / len(its) * 1e3, 4) for t in ts], "rtf": [round(audio / t, 1) for t in ts]}

    def sync_pass(its, seed):
        """The reference's own loop body, nothing pipelined (FastDiff.py:97-118): upload the mel, sample, normalise, bring the PCM back
        and WAIT, request by request -- a capture or a host stall is fully exposed here."""
        rows = infer._step_rows(model, N, None, None)
        dev = next(model.parameters()).device
        torch.cuda.synchronize()
        host_us = []
        t0 = time.perf_counter()
        with torch.no_grad():
            for i, it in enumerate(its):
                mel = it["mel"].t().contiguous().unsqueeze(0).to(dev)
                h0 = time.perf_counter()
                wav = model.sample(mel, rows, seed=seed, stream_ids=[i], defer_check=True)
                host_us.append((time.perf_counter() - h0) * 1e6)
                model.check()
                pcm = model.peak_normalize_int16(wav).cpu()
        dt = time.perf_counter() - t0
        assert pcm.shape == (1, its[-1]["len"] * HOP)
        host_us.sort()
        return dt, host_us[len(host_us) // 2]

    def sync_leg(its, passes, audio):
        rs = [sync_pass(its, 7 + k) for k in range(passes)]
        return {"ms_per_request": [round(t / len(its) * 1e3, 4) for t, _ in rs], "rtf": [round(audio / t, 1) for t, _ in rs],
                "host_us_in_sample_call_median": [round(h, 1) for _, h in rs]}

    res = {"requests": len(items), "frames_min_mean_max": [min(lens), t_mean, max(lens)], "distinct_lengths": len(set(lens)),
           "reverse_steps": N, "audio_s": round(audio_s, 2)}
    one_pass(fixed[:4], 1, 1, False)                      # the library's buffers and the pinned staging at their final size
    one_pass(sorted(items, key=lambda it: -it["len"])[:1], 1, 1, False)
    res["fixed_shape_control"] = leg(fixed, 1, False, 2, t_mean * len(fixed) * HOP / SR)
    res["b1"] = leg(items, 1, False, 2 if quick else 3, audio_s)
    res["b8"] = leg(items, 8, True, 2 if quick else 3, audio_s)
    res["b1_sync"] = sync_leg(items, 1 if quick else 2, audio_s)
    if quick:
        res["graph_cache"] = {k: model.counter(k) for k in ("graph_captures", "graph_hits", "graph_evictions", "graphs_resident")}
        res["all_legs"] = "python bench.py --workload stream"
        return res["b1"]["ms_per_request"][-1] * len(items) / 1e3, frames, res
    model.set_option("graph", "0")
    res["b1_no_graph"] = leg(items, 1, False, 2, audio_s)
    res["b8_no_graph"] = leg(items, 8, True, 2, audio_s)
    res["b1_sync_no_graph"] = sync_leg(items, 2, audio_s)
    model.set_option("graph", "1")
    model.set_option("t_bucket", "0")          # round 5's library: a graph per exact T, i.e. one capture per request
    res["b1_exact_T_graphs"] = leg(items, 1, False, 1, audio_s)
    res["b1_sync_exact_T_graphs"] = sync_leg(items, 1, audio_s)
    model.set_option("t_bucket", "32")
    model.set_option("graph", "0" if args.no_graph else "1")
    try:
        res["graph_cache"] = {k: model.counter(k) for k in ("graph_captures", "graph_hits", "graph_evictions", "graphs_resident")}
    except Exception as e:      # noqa: BLE001 -- a library without these counters
        res["graph_cache"] = {"error": repr(e)[:100]}
    # the headline of this workload: the LAST pass of the one-request-per-call leg (steady state of a long-running server / test set)
    elapsed = res["b1"]["ms_per_request"][-1] * len(items) / 1e3
    return elapsed, frames, res


def project_sharded(args, model, items, t_full, dev):
    """A PROJECTION, labelled as one (a gpurun box has one GPU): what a `project_ranks`-GPU run of this job would take, from pieces
    measured here.  t_share = the slowest of the R shares the LPT partition hands out, each run through the same per-rank code
    (infer.synthesize on device-resident mels, PCM left on the device); t_fixed = what rank 0 alone does around it (the collater's
    view of all utterances, one packed buffer per peer + its upload, the gathered job's one copy back to the host).  Not in it: the
    RCCL transport itself (<= 2.2 MB out / 3.5 MB back per peer over xGMI) and the start-up handshake."""
    import numpy as np
    from fastdiff_amd import infer, shard
    R, N, reps = args.project_ranks, args.nsteps, max(2, args.steps)
    lens = [it["len"] for it in items]
    parts = shard.partition_utterances(lens, R, cost="time" if args.balance == "time" else None)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r

    def prep():      # rank 0 (infer.synthesize_sharded -> shard.scatter_utterances): every mel as stored, packed once, one upload
        return shard.pack_messages([it["mel"] for it in items], parts, dev)
    t_prep, packed = timed(prep)
    shares = []
    for r, p in enumerate(parts):
        off, local = 0, []
        for i in p:
            local.append({"item_name": str(i), "mel": packed[r][off: off + 80 * lens[i]].view(lens[i], 80), "len": lens[i], "uid": i})
            off += 80 * lens[i]
        # gather = src: the share's PCM stays on the device (it leaves through the RCCL gather); none: it goes to this rank's host
        t_r, pcm = timed(lambda: infer.synthesize(model, local, N, args.batch, 1234, drop_last_frame=False, return_device=args.gather == "src"))
        shares.append(t_r)
    whole = torch.empty(sum(lens) * HOP, dtype=torch.int16, device=dev)
    t_back, _ = timed(lambda: whole.cpu())
    # the job's start-up message (names, noise-stream ids, lengths: pickle + two broadcasts + unpickle), timed on a gloo group of R
    # processes on this host -- a term of its own: against a per-rank share of ~10 ms it is not negligible
    bcast = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import broadcast_cost
        bcast = broadcast_cost.measure(world=R, n_items=len(items), reps=100)
    except Exception as e:      # noqa: BLE001
        bcast = {"error": repr(e)}
    t_bcast = (bcast.get("median_ms") or 0.0) * 1e-3
    if args.gather == "none":
        t_back = 0.0
    t_share, t_fixed = max(shares), t_prep + t_back + t_bcast
    return {"projected_ranks": R, "is_a_projection": True, "gather": args.gather, "balance": args.balance, "t_full_1gpu_ms": round(t_full * 1e3, 3),
            "t_share_ms": {"max": round(t_share * 1e3, 3), "min": round(min(shares) * 1e3, 3), "all": [round(s * 1e3, 3) for s in shares]},
            "t_fixed_rank0_ms": {"pack_and_upload": round(t_prep * 1e3, 3), "job_pcm_to_host": round(t_back * 1e3, 3),
                                 "startup_broadcast_object_list": round(t_bcast * 1e3, 3)},
            "startup_broadcast": bcast,
            "projected_ms_per_job": round((t_fixed + t_share) * 1e3, 3),
            "projected_efficiency": round(t_full / (R * (t_fixed + t_share)), 4),
            "not_included": "RCCL transport of <= 2.2 MB out / 3.5 MB back per peer over xGMI (the broadcast of names / ids / lengths is in, as timed on gloo)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="configs1", choices=("configs1", "config4", "config5", "stream"))
    ap.add_argument("--stream-requests", type=int, default=256, help="--workload stream: number of requests (all lengths distinct)")
    ap.add_argument("--batch", type=int, default=None, help="utterances per fd_sample call: default 8 (config4: micro-batch size, default 16; config5: 16)")
    ap.add_argument("--frames", type=int, default=864)
    ap.add_argument("--nsteps", type=int, default=None, help="reverse steps N (3,4,6,8,200,1000); default 4 (config4: 6)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-replay-profile", action="store_true", help="roofline from the in-process timestamped launches only (no rocprofv3 child run of the timed loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (cpu_baseline and parity)")
    ap.add_argument("--no-fp32-pipe", action="store_true")
    ap.add_argument("--no-b1", action="store_true", help="skip the one-utterance-per-call (reference CLI mode) object")
    ap.add_argument("--no-stream", action="store_true", help="skip the short stream object (256 requests of 256 different lengths, one per call) of the default line")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ragged", action="store_true",
                    help="BASELINE config 4 style batch: T_i ~ U{200..frames}, zero-padded; RTF counts the valid audio only")
    ap.add_argument("--no-lens", action="store_true", help="with --ragged: do not tell the library the lengths (padded compute)")
    ap.add_argument("--no-torch-eager-baseline", action="store_true",
                    help="skip the plain PyTorch-ROCm eager restatement of the same sampling on this GPU (3 repetitions, < 1 s)")
    ap.add_argument("--no-host-io", action="store_true", help="skip the extra host-to-host (PCIe-inclusive) measurement")
    ap.add_argument("--project-ranks", type=int, default=8, help="config4 on one GPU: also project the job onto this many ranks from measured shares (0 = off)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="library option (fd_set_option), repeatable")
    ap.add_argument("--weights", default=None, choices=("init", "contractive"),
                    help="init = FastDiff() default init, seed 1234 (BASELINE.md); contractive = the same with final_conv edited so that a long "
                         "schedule contracts like a trained model's (make_contractive_).  Default: init for N <= 8, contractive beyond")
    ap.add_argument("--gather", default="src", choices=("src", "none"),
                    help="config4: 'src' = the PCM of the whole job on rank 0's host (north_star's scatter / gather); 'none' = every rank keeps / "
                         "writes its own share, as the reference does (FastDiff.py:107-118)")
    ap.add_argument("--balance", default="time", choices=("time", "frames"), help="config4: what the LPT partition weighs (shard.utterance_cost | frames)")
    args = ap.parse_args()
    if args.nsteps is None:
        args.nsteps = 6 if args.workload == "config4" else 4
    if args.batch is None:      # (config4 on one GPU: micro-batches of 16 take 64.0 ms per job, of 8 68.1 ms: profiles/r05/s9_config4_microbatch.txt)
        args.batch = 16 if args.workload in ("config5", "config4") else 8
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if os.environ.get("FD_BENCH_OVERSUBSCRIBE") == "1" and "LOCAL_RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # ranks sharing one GPU get disjoint compute units (must be set before the ROCm runtime initialises): two vocoding processes on
        # the same CUs can disturb each other around process start / exit (profiles/r03/two_processes_one_gpu.txt)
        w_, r_ = int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
        per = 256 // w_
        os.environ.setdefault("HSA_CU_MASK", "0:%d-%d" % (r_ * per, (r_ + 1) * per - 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: fastdiff_amd has no CPU path")
    n_dev = torch.cuda.device_count()
    # FD_BENCH_OVERSUBSCRIBE=1 (tests of the multi-rank code path on a box with fewer GPUs): ranks share GPUs, the process group runs on
    # gloo, and the line says so ("n_gpus" stays the number of GPUs actually used, "oversubscribed": true) -- never a scaling number.
    oversub = os.environ.get("FD_BENCH_OVERSUBSCRIBE") == "1" and args.gpus > n_dev
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > n_dev and not oversub:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this box has {n_dev} GPU(s): refusing to print a line for a world that was not measured")
        if args.gpus > 1:
            sys.exit(self_spawn(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if local_rank >= n_dev and not oversub:
        raise SystemExit(f"bench.py: rank {rank} (local {local_rank}) has no GPU of its own: {n_dev} visible")
    gpu_index = local_rank % n_dev
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    placement = None
    if world > 1:      # one rank per GPU on a many-core host: each rank on its own slice of the cores next to its GPU (fastdiff_amd/affinity.py)
        try:
            from fastdiff_amd import affinity
            placement = affinity.bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        except Exception as e:      # noqa: BLE001 -- placement is best effort, never a reason not to measure
            placement = {"applied": False, "why": repr(e)[:100]}
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if oversub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            ones = torch.ones(1)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        assert rccl_ranks == world == dist.get_world_size(), (rccl_ranks, world)

    import fastdiff_amd
    from fastdiff_amd import sampler, schedules

    B, T, N = args.batch, args.frames, args.nsteps
    torch.manual_seed(1234)                       # weights = FastDiff() default init, seed 1234
    model = fastdiff_amd.FastDiff()
    if args.weights is None:
        args.weights = "contractive" if N > 8 else "init"
    if args.weights == "contractive":
        make_contractive_(model)
    model = model.to(dev).eval()
    if args.no_graph:
        model.set_option("graph", "0")
    for kv in args.opt:
        model.set_option(*kv.split("=", 1))
    global GEMM_FORM
    GEMM_FORM = model._options.get("gemm_form", "winograd")
    dh = schedules.training_hyperparams()
    rows = sampler.InferenceSchedule(dh, schedules.noise_schedule_for(N), verbose=False).rows()

    def barrier():
        if world > 1:
            dist.barrier() if oversub else dist.barrier(device_ids=[local_rank])

    lens = None
    if args.workload == "stream":
        if world != 1:
            raise SystemExit("bench.py: --workload stream is a one-GPU workload")
        elapsed, frames, projection = run_stream(args, model)
        args.steps = 1
        total_frames, padded_frames = frames, frames
        scaling = "weak"
        B, T = 1, projection["frames_min_mean_max"][2]
        workload = "reference CLI pattern: %d requests, 1 utterance per call, every T distinct in {200..864}, N=%d, host mel -> host PCM" % (projection["requests"], N)
    elif args.workload == "config5":
        if world != 1:
            raise SystemExit("bench.py: --workload config5 is BASELINE configs[4], a one-GPU configuration")
        elapsed, frames, projection = run_config5(args, model)
        total_frames, padded_frames = frames, projection["padded_frames"]
        scaling = "weak"
        workload = "BASELINE configs[4]: dir of 16 Tacotron-range [T,80] .npy mels -> collater -> 1 batch of 16, N=%d -> int16 PCM on host" % N
    elif args.workload == "config4":
        elapsed, frames, projection = run_config4(args, model, rank, world, local_rank, dev)
        total_frames, padded_frames = frames, frames
        scaling = "strong"
        workload = "BASELINE configs[3]: 64 ragged utts (T 200..864) on rank 0's host, N=%d, LPT scatter, micro-batches <=%d, %s" % (
            N, B, "int16 gather" if args.gather == "src" else "no gather (each rank keeps its PCM)")
    else:
        torch.manual_seed(1234 + rank)
        mel = (torch.rand(B, 80, T) * 7.5 - 6.0).to(dev)    # uniform on [mel_vmin, mel_vmax]
        valid_frames = B * T
        if args.ragged:
            lens = torch.randint(200, T + 1, (B,)).tolist() if T > 200 else [T] * B
            for b, t in enumerate(lens):
                mel[b, :, t:] = 0.0                          # collate_2d padding
            valid_frames = sum(lens)
        use_lens = None if args.no_lens else lens
        child = os.environ.get("FD_BENCH_CHILD") == "1"      # rocprof_graph_replay's child: the device-resident loop only
        host_metric = not (args.no_host_io or child)
        # SURVEY.md 8(d) defines the metric host to host: one step = pinned host mel -> device -> fd_sample (N reverse steps) -> int16
        # epilogue on the device -> pinned host PCM, calls pipelined one deep (the range check of call k is looked at after call k + 1
        # has been enqueued; a call that had to be redone gets its epilogue and copy again).  That is `value`.  The same K steps with the
        # mel resident in HBM and the waveform left there are timed right behind it: `value_device_resident`.
        mel_h = mel.cpu().pin_memory()
        pcm_h = torch.empty((B, T * HOP), dtype=torch.int16).pin_memory()
        valid = None if use_lens is None else [t * HOP for t in use_lens]

        def host_step(i):
            m = mel_h.to(dev, non_blocking=True)
            wav = model.sample(m, rows, seed=i, lens=use_lens, defer_check=True)
            pcm_h.copy_(model.peak_normalize_int16(wav, valid=valid), non_blocking=True)
            return model.last_ticket, wav

        def host_settle(prev):
            if prev is not None and model.settle(prev[0]):
                pcm_h.copy_(model.peak_normalize_int16(prev[1], valid=valid), non_blocking=True)

        def host_loop(k0, n):
            prev = None
            for i in range(n):
                cur = host_step(k0 + i)
                host_settle(prev)
                prev = cur
            host_settle(prev)

        def resident_loop(k0, n):
            out = None
            for i in range(n):
                out = model.sample(mel, rows, seed=k0 + i, lens=use_lens, defer_check=True)
            model.check()          # (every call but the last was looked at by its successor, the last one here)
            return out

        timed = host_loop if host_metric else resident_loop
        with torch.no_grad():
            out = model.sample(mel, rows, seed=0, lens=use_lens)
            timed(0, args.warmup)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            timed(100, args.steps)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            if child:
                return
            resident = None
            if host_metric:
                assert int(pcm_h.abs().max()) == 32767
                resident_loop(0, args.warmup)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                out = resident_loop(100, args.steps)
                torch.cuda.synchronize()
                resident = (time.perf_counter() - t1) / args.steps * 1e3
        assert torch.isfinite(out if lens is None else torch.stack([out[b, :, : lens[b] * HOP].abs().max() for b in range(B)])).all()
        total_frames = world * valid_frames        # (ragged: rank 0's draw stands for every rank)
        padded_frames = world * B * T
        scaling = "weak"
        workload = "BASELINE configs[%d]: LJSpeech shape, batch=%d x 80x%d mel per GPU, N=%d, HIP kernels + hipGraph sampler" % (2 if N > 8 else 1, B, T, N)
    t = torch.tensor([elapsed], dtype=torch.float64, device=None if oversub else dev)
    t_min = t.clone()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_min, op=dist.ReduceOp.MIN)      # a straggler shows as a gap between the two
    elapsed, elapsed_min = float(t.item()), float(t_min.item())
    ms_per_step = elapsed / args.steps * 1e3
    audio_s = total_frames * HOP / SR
    line = {
        "metric": "real-time factor (audio-sec/wall-sec), N=%d reverse steps, 80x%d mel" % (N, T),
        "value": round(audio_s / (ms_per_step / 1e3), 2),
        "unit": "x real-time",
        "samples_per_s": round(padded_frames * HOP / (ms_per_step / 1e3), 1),
        "n_gpus": (min(rccl_ranks, n_dev) if oversub else rccl_ranks) if world > 1 else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "ms_per_step_ranks": {"max": round(ms_per_step, 4), "min": round(elapsed_min / args.steps * 1e3, 4), "ranks": world},
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": workload,
                   "batch_per_gpu": B, "frames": T, "reverse_steps": N,
                   "sharding": ("utterances/rank, no data-path collective" if args.workload == "configs1" else
                                "one GPU" if args.workload in ("config5", "stream") else "LPT partition, one packed p2p message per peer each way (RCCL)"),
                   "value_is": (("host mel -> host int16 PCM (SURVEY 8d's definition); HBM-resident: value_device_resident" if not args.no_host_io
                                 else "HBM-resident mel -> HBM waveform (--no-host-io)") if args.workload == "configs1"
                                else "files of a directory -> host int16 PCM" if args.workload == "config5"
                                else "host mel -> host int16 PCM, last pass of the one-request-per-call leg" if args.workload == "stream" else "host mel -> host int16 PCM (rank 0)"),
                   "graph": not args.no_graph,
                   "weights": ("random init seed 1234 (no checkpoint offline)" if args.weights == "init" else
                               "random init seed 1234 + final_conv made contractive (x stays O(1) over a long schedule, as with a trained model)"),
                   "range_fallback": ("host-checked, pipelined (each call looked at after the next is enqueued, the last inside the timed region)"
                                      if model._options.get("fallback") == "host" else "in-graph fp32 twin behind every fp16x2 kernel"),
                   "world_size": world, "gpus_visible": n_dev, "oversubscribed": bool(oversub),
                   **({"gather": args.gather, "balance": args.balance} if args.workload == "config4" else {}),
                   **({"cpu_placement_rank0": placement} if placement is not None else {}),
                   **({"note": "ranks SHARE this box's GPU(s) on gloo with disjoint CU masks: a code-path proof, not a scaling number"} if oversub else {}),
                   "ragged": (None if not args.ragged else {"lens": lens, "told_to_library": not args.no_lens})},
    }
    if rank == 0 and args.workload == "configs1" and N > 8:
        # a schedule of more than 8 steps runs as 8-step pieces, each checked on the host; which pipe did the LAST timed call run on?
        line["long_schedule"] = {"pieces": model.counter("pieces"), "pieces_redone_on_fp32": model.counter("pieces_redone"),
                                 "pieces_enqueued_with_stages_on_fp32": model.counter("pieces_fp32"), "fp32_stage_mask": hex(model.counter("fp32_mask")),
                                 "of": "the last timed sample call (fd_get_counter); 0 / 0 = every piece ran on the default fp16x2 pipe"}
        line["config"]["parity_evidence"] = "all 1000 steps at T=864 vs the reference's float64 run, every 125th state (tests: test_config3_n1000_at_864_frames_against...)"
    if rank == 0:
        line["box"] = box_state()
        if args.workload == "config4" and projection is not None:
            line["projection" if world == 1 else "sharded_job_check"] = projection
        if args.workload == "config5":
            line["config5"] = projection
        if args.workload == "stream":
            line["stream"] = projection
            line["stream_b1_ms"], line["stream_b1_rtf"] = projection["b1"]["ms_per_request"][-1], projection["b1"]["rtf"][-1]
            line["stream_b1_first_pass_ms"] = projection["b1"]["ms_per_request"][0]
            line["stream_b8_ms"], line["stream_b1_no_graph_ms"] = projection["b8"]["ms_per_request"][-1], projection["b1_no_graph"]["ms_per_request"][-1]
            line["stream_fixed_shape_ms"] = projection["fixed_shape_control"]["ms_per_request"][-1]
            line["stream_b1_sync_ms"] = projection["b1_sync"]["ms_per_request"][-1]
            line["stream_b1_sync_exact_T_graphs_ms"] = projection["b1_sync_exact_T_graphs"]["ms_per_request"][-1]
            line["stream_b1_exact_T_graphs_ms"] = projection["b1_exact_T_graphs"]["ms_per_request"][-1]
    if rank == 0 and args.workload == "configs1" and resident is not None:
        # (this rank's own loop: no barrier around it)
        line["value_device_resident"] = round(audio_s / (resident / 1e3), 2)      # (world > 1: rank 0's loop standing for every rank)
        line["ms_per_step_device_resident"] = round(resident, 4)
        line["value_host_to_host"], line["ms_per_step_host_to_host"] = line["value"], line["ms_per_step"]
    if rank == 0 and world == 1 and args.workload == "configs1":
        use_lens = None if args.no_lens else lens
        if not args.no_host_io:
            line["host_inclusive"] = host_inclusive(model, mel, rows, use_lens, audio_s, reps=args.steps)      # a second run of the same loop + the PCIe legs on their own
        if not args.no_roofline:
            replay = None if (args.no_replay_profile or args.no_graph or args.ragged) else rocprof_graph_replay(args)
            roof, table = measure_roofline(model, mel, rows, B, T, N, use_lens, replay)
            line["roofline"] = roof
            line["kernels"] = table
        if B > 1 and not args.ragged and not args.no_b1:
            try:
                line["b1"] = b1_object(model, mel, rows, args.steps)
            except Exception as e:      # noqa: BLE001
                line["b1"] = {"error": repr(e)}
        if N <= 8 and not args.ragged and not args.no_stream and not args.no_graph:
            # the reference CLI's own call pattern (one utterance per call, a new length every call): the short form of --workload stream
            try:
                _, _, st = run_stream(args, model, quick=True)
                line["stream"] = st
                line["stream_b1_ms"], line["stream_b1_rtf"] = st["b1"]["ms_per_request"][-1], st["b1"]["rtf"][-1]
                line["stream_b1_sync_ms"], line["stream_b8_ms"] = st["b1_sync"]["ms_per_request"][-1], st["b8"]["ms_per_request"][-1]
                line["stream_fixed_shape_ms"] = st["fixed_shape_control"]["ms_per_request"][-1]
            except Exception as e:      # noqa: BLE001
                line["stream"] = {"error": repr(e)}
        if not args.no_fp32_pipe:
            try:
                line["fp32_pipe"] = fp32_pipe(model, mel, rows, use_lens, audio_s)
            except Exception as e:      # noqa: BLE001
                line["fp32_pipe"] = {"error": repr(e)}
        if not args.no_torch_eager_baseline:
            try:
                line["torch_eager_baseline"] = torch_eager_baseline(mel, rows, audio_s)
            except Exception as e:      # noqa: BLE001
                line["torch_eager_baseline"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            try:
                line["parity"] = parity_vs_oracle(T, rows, dev)
            except Exception as e:      # noqa: BLE001 -- the checker is optional for the measurement
                line["parity"] = {"error": repr(e)}
            try:
                line["cpu_baseline"] = cpu_baseline(T, rows)
            except Exception as e:      # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        # short scalars, top level AND inside the objects the driver's record keeps (it drops nested objects and cuts strings at
        # ~128 characters); `summary` goes last so that the tail of the line holds it
        summ = {"value": line["value"], "ms_per_step": line["ms_per_step"]}
        if "value_device_resident" in line:
            summ["value_device_resident"], summ["ms_device_resident"] = line["value_device_resident"], line["ms_per_step_device_resident"]
        if "value_host_to_host" in line:
            summ["value_host_to_host"], summ["ms_host_to_host"] = line["value_host_to_host"], line["ms_per_step_host_to_host"]
        if isinstance(line.get("b1"), dict) and "ms_per_step" in line["b1"]:
            summ["b1_ms"], summ["b1_rtf"] = line["b1"]["ms_per_step"], line["b1"]["value"]
            summ["b1_host_us"] = line["b1"].get("host_us_per_call_median")
            summ["b1_lvc12_frac_min_bytes"] = line["b1"].get("lvc_all_12_launches_frac_minimal_bytes")
        for k in ("stream_b1_ms", "stream_b1_rtf", "stream_b1_sync_ms", "stream_b8_ms", "stream_fixed_shape_ms"):
            if k in line:
                summ[k] = line[k]
        if isinstance(line.get("fp32_pipe"), dict) and "ms_per_step" in line["fp32_pipe"]:
            summ["fp32_pipe_ms"], summ["fp32_pipe_rtf"] = line["fp32_pipe"]["ms_per_step"], line["fp32_pipe"]["value"]
        if isinstance(line.get("parity"), dict) and "max_abs_diff_f16x2" in line["parity"]:
            summ["parity_f16x2"], summ["parity_fp32"] = line["parity"]["max_abs_diff_f16x2"], line["parity"]["max_abs_diff_fp32"]
        if isinstance(line.get("torch_eager_baseline"), dict) and "ms_per_step" in line["torch_eager_baseline"]:
            summ["torch_eager_gpu_ms"] = line["torch_eager_baseline"]["ms_per_step"]
            summ["torch_eager_like_reference_gpu_ms"] = line["torch_eager_baseline"].get("like_the_reference", {}).get("ms_per_step")
        roof = line.get("roofline") or {}
        for k in ("frac", "avg_launch_us", "lvc12_frac_min_bytes", "lvc12_frac_survey_bytes", "frac_h8", "frac_h64", "frac_up_h64", "frac_up_h256",
                  "frac_final_h256", "frac8d_up_h64", "frac8d_up_h256", "frac8d_final_h256", "gemm_us", "gemm_hbm_frac", "gemm_tflops_executed"):
            if roof.get(k) is not None:
                summ[("roofline_" + k) if k in ("frac", "avg_launch_us") else k] = roof[k]
        cb = line.get("cpu_baseline") or {}
        for k, v in cb.items():
            if k.startswith("rtf_"):
                summ["cpu_" + k] = v
        for k, v in summ.items():
            if k not in line:
                line[k] = v
        for k in ("b1_ms", "stream_b1_ms", "fp32_pipe_ms", "parity_f16x2", "parity_fp32", "value_host_to_host", "torch_eager_gpu_ms"):
            if k in summ:
                line["config"][k] = summ[k]

        def clip(o):      # strings of the objects the driver keeps: <= 120 characters (its own cut would lose the end silently)
            for k, v in list(o.items()):
                if isinstance(v, str) and len(v) > 120:
                    o[k] = v[:117] + "..."
        for o in (line, line["config"], line.get("roofline") or {}, line.get("cpu_baseline") or {}):
            clip(o)
        line["summary"] = summ
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
