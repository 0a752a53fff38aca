"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU oracle
and the reference-generated golden fixtures.

Tolerances: the reference's own fp32-vs-fp64 noise floor for one forward is ~3e-6 on
|y| <= 8, so  single forward  max|d| <= 2e-5;   N<=8 loop with injected noise  max|d| <= 1e-4;
N=1000 loop: within 10x of the fp32 reference's own drift against its fp64 run.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-5
LOOP_TOL = 1e-4


@pytest.fixture(scope="module")
def gc():
    import gpu_common
    return gpu_common


@pytest.fixture(scope="module")
def model(gc):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return gc.make_model()


@pytest.fixture(scope="module")
def sched():
    return load_golden("schedule")


def test_native_library_is_the_one_running(model):
    """The ops must come from the in-tree HIP library, not a fallback."""
    from fastdiff_amd import _capi
    lib = _capi.load()
    assert lib.fd_version().startswith(b"fastdiff_hip")
    maps = open("/proc/self/maps").read()
    assert "fastdiff_amd/lib/libfastdiff_hip.so" in maps


# ------------------------------------------------------------------------------------------------ forward vs golden
@pytest.mark.parametrize("case", ["f1", "f2", "f3", "f4"])
def test_forward_matches_reference_golden(model, gc, case):
    g = load_golden("forward_" + case)
    model.set_option("kernels", "fast")
    y = gc.run_forward(model, g["audio"], g["mel"], g["steps"])
    assert gc.maxdiff(y, g["y_f64"]) < FWD_TOL
    assert gc.maxdiff(y, g["y_f32"]) < FWD_TOL


def test_forward_naive_kernels_match_golden(model, gc):
    g = load_golden("forward_f2")
    model.set_option("kernels", "naive")
    try:
        y = gc.run_forward(model, g["audio"], g["mel"], g["steps"])
    finally:
        model.set_option("kernels", "fast")
    assert gc.maxdiff(y, g["y_f64"]) < FWD_TOL


def test_every_intermediate_matches_reference(model, gc):
    """Per-function pins: first conv, the three DBlocks, each KernelPredictor's kernels/bias,
    each LVC block output -- against the reference's own forward hooks."""
    g = load_golden("forward_f1")
    model.set_option("kernels", "fast")
    model.set_option("taps", "1")
    try:
        y = gc.run_forward(model, g["audio"], g["mel"], g["steps"])
        taps = gc.read_taps(model, 1, 4)
    finally:
        model.set_option("taps", "0")
    for k in ("a0", "a1", "a2", "a3", "x0", "x1", "x2"):
        assert gc.maxdiff(taps[k], g["tap_" + k]) < FWD_TOL, k
    for n in range(3):
        assert gc.maxdiff(taps[f"kernels{n}"], g[f"tap_kp{n}_kernels"]) < FWD_TOL, n
        assert gc.maxdiff(taps[f"bias{n}"], g[f"tap_kp{n}_bias"]) < FWD_TOL, n
    assert gc.maxdiff(y, g["y_f64"]) < FWD_TOL


@pytest.mark.parametrize("stage", ["first", "dblock", "kp_front", "kp_gemm", "convt", "lvc", "final"])
def test_each_fast_stage_against_oracle(model, gc, oracle64, stage):
    """One fast stage at a time on top of the naive set: localises a wrong kernel."""
    import synth
    B, T = 2, 37                      # odd T: partial tiles in every kernel
    mel, audio = synth.synth_mel(7, B, T), synth.synth_audio(7, B, T)
    steps = np.array([3.25, 480.5], np.float32)
    y_ref = oracle64.forward(audio, mel, steps)
    model.set_option("kernels", "naive")
    model.set_option("kernels." + stage, "fast")
    try:
        y = gc.run_forward(model, audio, mel, steps)
    finally:
        model.set_option("kernels", "fast")
    assert gc.maxdiff(y, y_ref) < FWD_TOL


def _predicted_kernel_error(model, gc, oracle64, audio, mel, steps, B, T):
    """max |predicted kernels / biases - float64 oracle| over the three KernelPredictors, and the eps error."""
    y_ref, ref = oracle64.forward(audio, mel, steps, taps=True)
    model.set_option("taps", "1")
    try:
        y = gc.run_forward(model, audio, mel, steps)
        taps = gc.read_taps(model, B, T)
    finally:
        model.set_option("taps", "0")
    errs = [gc.maxdiff(taps[f"kernels{n}"], ref[f"kernels{n}"]) for n in range(3)]
    errs += [gc.maxdiff(taps[f"bias{n}"], ref[f"bias{n}"]) for n in range(3)]
    return max(errs), gc.maxdiff(y, y_ref)


def test_predictor_gemm_f16x2_is_no_worse_than_fp32_pipe(model, gc, oracle64):
    """The default predictor GEMM runs on the fp16 matrix pipe with 2-piece (22-bit) operands; the fp32-MFMA form is kept
    behind option gemm=fp32.  Both must sit inside the forward tolerance, and the split form must not be the less accurate
    one (it accumulates the small cross terms separately)."""
    import synth
    B, T = 2, 130                     # 2 full 64-frame items + a ragged one per utterance
    mel, audio = synth.synth_mel(21, B, T), synth.synth_audio(21, B, T)
    steps = np.array([1.0, 733.5], np.float32)
    err = {}
    try:
        for mode in ("f16x2", "fp32"):
            model.set_option("gemm", mode)
            err[mode] = _predicted_kernel_error(model, gc, oracle64, audio, mel, steps, B, T)
    finally:
        model.set_option("gemm", "f16x2")
    print("predicted-kernel / eps max error vs float64 oracle:", err)
    for mode in err:
        assert err[mode][0] < FWD_TOL and err[mode][1] < FWD_TOL, (mode, err[mode])
    assert err["f16x2"][0] <= 1.5 * err["fp32"][0]


def test_predictor_gemm_out_of_fp16_range_falls_back_on_device(gc, oracle64):
    """|h| >= 32768 cannot be split into fp16 pieces: k_h_split flags it and the fp32 kernel behind the fp16 one does the
    step (include/fastdiff_hip.h, option "gemm").  Forced here by scaling the last predictor residual conv of block 0."""
    import synth
    sd = synth.synth_state_dict(1234)
    key = [k for k in sd if k.startswith("lvc_blocks.0.kernel_predictor.residual_conv") and k.endswith("weight_g")][-1]
    sd = dict(sd)
    sd[key] = (sd[key] * 3.0e5).astype(np.float32)
    m = gc.fastdiff_amd.FastDiff()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    o = type(oracle64)("f64")
    o.set_weights(sd)
    B, T = 1, 70
    mel, audio = synth.synth_mel(5, B, T), synth.synth_audio(5, B, T)
    steps = np.array([12.0], np.float32)
    y_ref, ref = o.forward(audio, mel, steps, taps=True)
    m.set_option("taps", "1")
    y = gc.run_forward(m, audio, mel, steps)
    taps = gc.read_taps(m, B, T)
    scale = float(np.abs(ref["kernels0"]).max())
    assert scale > 1.0e3                                   # the operands really were out of range
    assert np.isfinite(y).all()
    assert gc.maxdiff(taps["kernels0"], ref["kernels0"]) < 1e-5 * scale
    assert gc.maxdiff(taps["kernels1"], ref["kernels1"]) < FWD_TOL      # blocks 1, 2 unaffected


def test_lvc_f16x2_against_fp32_pipe_and_oracle(model, gc, oracle64):
    """LVC layers of hop 64 / 256 run on the fp16 matrix pipe with 2-piece operands by default; option lvc=fp32 keeps the
    fp32-MFMA kernels.  Every block output and eps must sit inside the forward tolerance in both modes."""
    import synth
    B, T = 2, 9                       # hop 64: 2.25 workgroup tiles (ragged); hop 256: 9 tiles
    mel, audio = synth.synth_mel(31, B, T), synth.synth_audio(31, B, T)
    steps = np.array([0.0, 999.0], np.float32)
    y_ref, ref = oracle64.forward(audio, mel, steps, taps=True)
    err = {}
    try:
        for mode in ("f16x2", "fp32"):
            model.set_option("lvc", mode)
            model.set_option("taps", "1")
            y = gc.run_forward(model, audio, mel, steps)
            taps = gc.read_taps(model, B, T)
            assert not model.read_tap("range_flags").view(np.int32)[:13].any()
            err[mode] = [gc.maxdiff(taps[k], ref[k]) for k in ("x0", "x1", "x2")] + [gc.maxdiff(y, y_ref)]
    finally:
        model.set_option("lvc", "f16x2")
        model.set_option("taps", "0")
    print("x0, x1, x2, eps max error vs float64 oracle:", err)
    for mode in err:
        assert max(err[mode]) < FWD_TOL, (mode, err[mode])


def test_lvc_out_of_fp16_range_falls_back_on_device(gc, oracle64):
    """|x + skip| >= 32768 in an LVC layer: k_lvc_h2 raises the layer's flag and the fp32 kernel launched behind it redoes
    the layer.  Forced by a huge upsampler in block 2 (hop 256); blocks 0 and 1 stay on the fp16 pipe."""
    import synth
    sd = dict(synth.synth_state_dict(1234))
    sd["lvc_blocks.2.upsample.weight"] = (sd["lvc_blocks.2.upsample.weight"] * 4.0e6).astype(np.float32)
    m = gc.fastdiff_amd.FastDiff()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    o = type(oracle64)("f64")
    o.set_weights(sd)
    B, T = 1, 5
    mel, audio = synth.synth_mel(6, B, T), synth.synth_audio(6, B, T)
    steps = np.array([77.0], np.float32)
    y_ref, ref = o.forward(audio, mel, steps, taps=True)
    m.set_option("taps", "1")
    y = gc.run_forward(m, audio, mel, steps)
    taps = gc.read_taps(m, B, T)
    flags = m.read_tap("range_flags").view(np.int32)
    assert not flags[:9].any() and flags[9:13].all(), flags[:13]        # [0] GEMM, [1..4] block 0, [5..8] block 1, [9..12] block 2
    scale = float(np.abs(ref["x2"]).max())
    assert scale > 32768.0
    assert gc.maxdiff(taps["x1"], ref["x1"]) < FWD_TOL
    assert gc.maxdiff(taps["x2"], ref["x2"]) < 2e-6 * scale
    assert np.isfinite(y).all() and gc.maxdiff(y, y_ref) < 2e-6 * max(1.0, float(np.abs(y_ref).max()))


def test_final_conv_fused_into_the_last_lvc_layer(model, gc, oracle64):
    """By default the last LVC layer applies final_conv to its own tile and leaves the sums in a zeroed accumulator (two atomic
    addends per tile-edge word): against the oracle and the unfused pair of kernels (option fuse_final=0); bit-reproducible; an
    operand outside the fp16 range in that layer falls back to the fp32 layer + the plain final_conv, and the accumulator is
    clean again for the next call."""
    import synth
    B, T = 2, 9                                             # 9 tiles of 256 columns per utterance: 8 inner tile edges each
    mel, audio = synth.synth_mel(44, B, T), synth.synth_audio(44, B, T)
    steps = np.array([12.5, 803.0], np.float32)
    y_ref = oracle64.forward(audio, mel, steps)
    y_f = gc.run_forward(model, audio, mel, steps)
    assert not model.read_tap("range_flags").view(np.int32).any()
    try:
        model.set_option("fuse_final", "0")
        y_u = gc.run_forward(model, audio, mel, steps)
    finally:
        model.set_option("fuse_final", "1")
    print("eps max error vs float64 oracle: fused %.2e, unfused %.2e" % (gc.maxdiff(y_f, y_ref), gc.maxdiff(y_u, y_ref)))
    assert gc.maxdiff(y_f, y_ref) < FWD_TOL and gc.maxdiff(y_u, y_ref) < FWD_TOL
    assert np.array_equal(gc.run_forward(model, audio, mel, steps), y_f)
    big = (audio * 3.0e7).astype(np.float32)                # drives every activation of the audio path past 32768
    y_big_ref = oracle64.forward(big, mel, steps)
    y_big = gc.run_forward(model, big, mel, steps)
    flags = model.read_tap("range_flags").view(np.int32)
    assert flags[12] and flags[0] == 0, flags[:20]
    assert np.isfinite(y_big).all() and gc.maxdiff(y_big, y_big_ref) < 3e-6 * float(np.abs(y_big_ref).max())
    assert np.array_equal(gc.run_forward(model, audio, mel, steps), y_f)


def test_conv_f16x2_against_fp32_pipe_and_oracle(model, gc, oracle64):
    """DBlocks and ConvTranspose upsamplers: fp16 pipe with 2-piece operands (default) vs option conv=fp32."""
    import synth
    B, T = 2, 33
    mel, audio = synth.synth_mel(41, B, T), synth.synth_audio(41, B, T)
    steps = np.array([5.0, 640.25], np.float32)
    y_ref, ref = oracle64.forward(audio, mel, steps, taps=True)
    err = {}
    try:
        for mode in ("f16x2", "fp32"):
            model.set_option("conv", mode)
            model.set_option("taps", "1")
            y = gc.run_forward(model, audio, mel, steps)
            taps = gc.read_taps(model, B, T)
            assert not model.read_tap("range_flags").view(np.int32).any()
            err[mode] = [gc.maxdiff(taps[k], ref[k]) for k in ("a1", "a2", "a3", "x0", "x1", "x2")] + [gc.maxdiff(y, y_ref)]
    finally:
        model.set_option("conv", "f16x2")
        model.set_option("taps", "0")
    print("a1, a2, a3, x0, x1, x2, eps max error vs float64 oracle:", err)
    for mode in err:
        assert max(err[mode]) < FWD_TOL, (mode, err[mode])


def test_everything_out_of_fp16_range_falls_back_on_device(gc, oracle64):
    """A first conv scaled by 1e7 drives every activation past 32768: every fp16-pipe launch that sees them (DBlocks,
    ConvTranspose, LVC layers) must raise its flag and be redone by the fp32 kernel behind it.  The result is compared
    relative to its own scale."""
    import synth
    sd = dict(synth.synth_state_dict(1234))
    sd["first_audio_conv.weight_g"] = (sd["first_audio_conv.weight_g"] * 1.0e7).astype(np.float32)
    m = gc.fastdiff_amd.FastDiff()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    o = type(oracle64)("f64")
    o.set_weights(sd)
    B, T = 1, 7
    mel, audio = synth.synth_mel(8, B, T), synth.synth_audio(8, B, T)
    steps = np.array([300.0], np.float32)
    y_ref, ref = o.forward(audio, mel, steps, taps=True)
    m.set_option("taps", "1")
    y = gc.run_forward(m, audio, mel, steps)
    taps = gc.read_taps(m, B, T)
    flags = m.read_tap("range_flags").view(np.int32)
    assert flags[0] == 0                                     # the predictor sees only the mel
    assert flags[13:16].all() and flags[16:19].all()         # DBlocks, ConvTranspose
    assert flags[1:13].reshape(3, 4)[1:].all()               # LVC layers of hop 64 and 256 (hop 8 has no fp16 kernel)
    assert np.isfinite(y).all()
    for k in ("a1", "a2", "a3", "x0", "x1", "x2"):
        scale = float(np.abs(ref[k]).max())
        assert scale > 32768.0, k
        assert gc.maxdiff(taps[k], ref[k]) < 3e-6 * scale, k
    assert gc.maxdiff(y, y_ref) < 3e-6 * float(np.abs(y_ref).max())


def test_sampler_hands_over_to_fp32_at_64_frames(gc, sched):
    """The hand-over inside the sampler at a length with tile edges (64 frames, a ragged batch of two): with the first conv scaled by
    3e4 (x grows by about that factor per reverse step: four steps stay finite in float32) every DBlock / ConvTranspose / LVC launch
    of step 0 raises its flag and is redone by the fp32 kernel behind it; from step 1
    on those launches skip their fp16 attempt (skip_after_previous_overflow).  The result must agree with the same call on the
    fp32 kernels selected outright (options lvc = conv = fp32), relative to its own scale, and the call's flags must say so."""
    import synth
    sd = dict(synth.synth_state_dict(1234))
    sd["first_audio_conv.weight_g"] = (sd["first_audio_conv.weight_g"] * 3.0e4).astype(np.float32)
    B, T, N = 2, 64, 4
    lens = [64, 40]
    rows, _ = gc.table_rows(sched, N)
    mel = torch.from_numpy(synth.synth_mel(9, B, T)).cuda()
    mel[1, :, lens[1]:] = 0.0
    out = {}
    for tag, opts in (("handover", {"fallback": "graph"}), ("hostcheck", {"fallback": "host"}), ("fp32", {"lvc": "fp32", "conv": "fp32"})):
        m = gc.fastdiff_amd.FastDiff()
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        m = m.cuda().eval()
        for k, v in opts.items():
            m.set_option(k, v)
        with torch.no_grad():
            y = m.sample(mel, rows, seed=5, lens=lens, stream_ids=[0, 1])
        out[tag] = [y[b, 0, : lens[b] * 256].cpu().numpy() for b in range(B)]
        flags = m.read_tap("range_flags_call").view(np.int32)
        if tag == "handover":
            assert flags[0] == 0 and flags[13:19].all() and flags[1:13].reshape(3, 4)[1:].all(), flags[:20]
        elif tag == "fp32":
            assert not flags[1:19].any()
        # (host check: the flags left behind are those of the second pass -- stages that met garbage, not large values, in the first
        # pass and so were flagged only now, with their fp32 twin inline -- nothing to assert on them but the result below)
    for b in range(B):
        scale = float(np.abs(out["fp32"][b]).max())
        assert scale > 32768.0
        for tag in ("handover", "hostcheck"):
            assert np.isfinite(out[tag][b]).all() and gc.maxdiff(out[tag][b], out["fp32"][b]) < 1e-5 * scale, (tag, b, scale)


def test_ragged_batch_with_lens_equals_each_utterance_alone(gc, sched):
    """BASELINE config 4 in small: utterances of different length in one zero-padded batch.  With `lens` every utterance must
    come out bit-identical to running it alone at its own length (forward, and the N-step sampler through the cached graph),
    whatever stale data the skipped regions of the workspace hold; without `lens` the padded tensor is computed as is."""
    import synth
    m = gc.make_model()
    B, T = 4, 300
    lens = [300, 37, 150, 1]
    mel = synth.synth_mel(51, B, T)
    audio = synth.synth_audio(51, B, T)
    for b, t in enumerate(lens):
        mel[b, :, t:] = 0.0                                   # collate_2d padding
    steps = np.array([3.0, 77.5, 500.0, 998.0], np.float32)
    rows, _ = gc.table_rows(sched, 4)
    N = len(rows)
    x_T = synth.hash_normal(9, 1, B * T * 256).reshape(B, 1, T * 256)
    z = np.stack([synth.hash_normal(9, 2 + k, B * T * 256).reshape(B, 1, T * 256) for k in range(N)])
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    with torch.no_grad():
        alone, alone_s = [], []
        for b, t in enumerate(lens):
            alone.append(m((cu(audio[b:b + 1, :, : t * 256]), cu(mel[b:b + 1, :, :t]), cu(steps[b:b + 1].reshape(1, 1)))))
            alone_s.append(m.sample(cu(mel[b:b + 1, :, :t]), rows, x_T=cu(x_T[b:b + 1, :, : t * 256]), noise=cu(z[:, b:b + 1, :, : t * 256])))
        # leave different stale data in every workspace buffer, then the ragged calls
        m((cu(synth.synth_audio(52, B, T)), cu(synth.synth_mel(52, B, T)), cu(steps.reshape(-1, 1))))
        part = m((cu(audio), cu(mel), cu(steps.reshape(-1, 1))), lens=lens)
        m.sample(cu(synth.synth_mel(53, B, T)), rows, seed=6)
        part_s = m.sample(cu(mel), rows, x_T=cu(x_T), noise=cu(z), lens=lens)
        part_s2 = m.sample(cu(mel), rows, x_T=cu(x_T), noise=cu(z), lens=lens)        # cached graph
        full = m((cu(audio), cu(mel), cu(steps.reshape(-1, 1))))                        # no lens: the padded tensor
    for b, t in enumerate(lens):
        n = t * 256
        assert torch.equal(part[b, :, :n], alone[b][0]), b
        assert torch.equal(part_s[b, :, :n], alone_s[b][0]), b
        assert torch.equal(part_s2[b, :, :n], alone_s[b][0]), b
    assert torch.equal(full[0], alone[0][0])                                             # the full-length item is the same either way
    assert not torch.equal(full[1, :, : 37 * 256], alone[1][0])                          # a padded item differs near its end: that is the reference's padded result
    assert not m.read_tap("range_flags").view(np.int32).any()
    with pytest.raises(Exception, match="lens"):
        m((cu(audio), cu(mel), cu(steps.reshape(-1, 1))), lens=[300, 0, 150, 1])


def test_forward_on_the_mel_of_a_real_recording(model, gc, oracle64):
    """SURVEY 8c pin (4): a realistic conditioning -- the log-mel of the reference's sample recording LJ001-0002 (163 frames
    after the collater drops the last one), whose statistics differ from the uniform synthetic mels (silences at -4.8, peaks
    at +0.6) -- through the whole denoiser against the float64 oracle."""
    import synth
    g = load_golden("frontend_lj001_0002")
    mel = np.ascontiguousarray(g["mel_f64"][None, :, :163]).astype(np.float32)
    audio = synth.synth_audio(77, 1, 163)
    steps = np.array([498.0537], np.float32)                      # the N=4 schedule's first mapped step
    y_ref = oracle64.forward(audio, mel, steps)
    y = gc.run_forward(model, audio, mel, steps)
    assert gc.maxdiff(y, y_ref) < FWD_TOL
    assert not model.read_tap("range_flags").view(np.int32).any()


@pytest.mark.parametrize("B,T", [(1, 1), (3, 3), (1, 63), (2, 130)])
def test_forward_ragged_sizes_against_oracle(model, gc, oracle64, B, T):
    import synth
    mel, audio = synth.synth_mel(11 + T, B, T), synth.synth_audio(11 + T, B, T)
    steps = np.linspace(0.0, 999.0, B).astype(np.float32)
    y_ref = oracle64.forward(audio, mel, steps)
    y = gc.run_forward(model, audio, mel, steps)
    assert gc.maxdiff(y, y_ref) < FWD_TOL


def test_random_shapes_default_pipe_against_fp32_pipe(gc):
    """A sweep over shapes nobody picked by hand: 64 seeded draws of (B, T, lens, N) -- one frame, tile edges (T * hop around 64,
    128, 256 columns), ragged batches, schedules that are one graph, whole 8-step pieces and pieces with a remainder, with and
    without the hoisted predictor -- each sampled on the default pipe (2-piece fp16 operands, hoisting, host-checked range) and on
    the exact-fp32 kernels (themselves held against the float64 oracle above), same seed and noise.  Bar: the N-step loop tolerance."""
    import synth
    rng = np.random.RandomState(20240917)
    fast, exact = gc.make_model(), gc.make_model()
    for k in ("gemm", "lvc", "conv"):
        exact.set_option(k, "fp32")
    exact.set_option("hoist", "off")
    worst = 0.0
    with torch.no_grad():
        for draw in range(64):
            B = int(rng.choice([1, 1, 2, 3, 5, 8, 9]))
            T = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 48, 65, 100, 130, 257]))
            N = int(rng.choice([1, 2, 3, 4, 6, 8, 9, 11, 16, 17]))
            lens = [int(v) for v in rng.randint(1, T + 1, size=B)]
            lens[int(rng.randint(B))] = T
            use_lens = lens if (B > 1 and draw % 3) else None
            mel = torch.from_numpy(synth.synth_mel(900 + draw, B, T)).cuda()
            if use_lens:
                for b, t in enumerate(use_lens):
                    mel[b, :, t:] = 0.0
            rows = [{"t": float(rng.uniform(0, 999)), "c_eps": 0.03, "c_div": 0.995, "sigma": 0.1, "c1": 1.0, "c2": 0.0, "c3": 0.0,
                     "add_noise": int(k < N - 1)} for k in range(N)]
            fast.set_option("hoist", "off" if draw % 4 == 3 else "auto")
            a = fast.sample(mel, rows, seed=draw, lens=use_lens)
            b = exact.sample(mel, rows, seed=draw, lens=use_lens)
            assert torch.isfinite(a).all() and torch.isfinite(b).all(), (draw, B, T, N, use_lens)
            for i in range(B):
                n = (use_lens[i] if use_lens else T) * 256
                d = float((a[i, :, :n] - b[i, :, :n]).abs().max())
                assert d <= LOOP_TOL * max(1.0, float(b[i, :, :n].abs().max())), (draw, B, T, N, use_lens, i, d)
                worst = max(worst, d)
    assert worst > 0.0          # (two different arithmetic pipes: identical bits would mean the option did not take)


def test_forward_accepts_unbatched_mel_and_long_steps(model, gc):
    """egs/demo.ipynb passes a [80,T] mel; training passes integer steps (util.py:311-319)."""
    g = load_golden("forward_f1")
    with torch.no_grad():
        y = model((torch.from_numpy(g["audio"]).cuda(), torch.from_numpy(g["mel"][0]).cuda(),
                   torch.tensor([[7]], dtype=torch.long).cuda()))
        y2 = model((torch.from_numpy(g["audio"]).cuda(), torch.from_numpy(g["mel"]).cuda(), torch.tensor([[7.0]]).cuda()))
    assert torch.equal(y, y2)


def test_length_mismatch_is_the_references_assert(model):
    with pytest.raises(AssertionError, match="not matched"):        # modules.py:236
        model((torch.zeros(1, 1, 255).cuda(), torch.zeros(1, 80, 1).cuda(), torch.zeros(1, 1).cuda()))


def test_remove_weight_norm_changes_nothing(gc):
    """a11: folding weight-norm on load == evaluating it every forward (FastDiff_model.py:104-122)."""
    g = load_golden("forward_f1")
    m = gc.make_model()
    y0 = gc.run_forward(m, g["audio"], g["mel"], g["steps"])
    m.remove_weight_norm()
    y1 = gc.run_forward(m, g["audio"], g["mel"], g["steps"])
    assert gc.maxdiff(y0, y1) < 1e-5   # the library folds with a double-precision norm, torch in fp32


# ------------------------------------------------------------------------------------------------ sampler vs golden
@pytest.mark.parametrize("case", ["s1", "s2", "s3", "s5"])
def test_sampler_matches_reference_golden(model, gc, sched, case):
    g = load_golden("sample_" + case)
    N, ddim = int(g["N"]), bool(g["ddim"])
    B, _, T = g["mel"].shape
    rows, _ = gc.table_rows(sched, N)
    noise = gc.exec_order_noise(gc.noise_from_seed(int(g["seed"]), B, T, N))
    seq = "seq_f64" in g
    with torch.no_grad():
        res = model.sample(torch.from_numpy(g["mel"]).cuda(), rows, ddim=ddim, x_T=torch.from_numpy(g["x_T"]).cuda(),
                           noise=torch.from_numpy(noise).cuda(), return_sequence=seq)
    torch.cuda.synchronize()
    got = np.stack([r.cpu().numpy() for r in res]) if seq else res.cpu().numpy()
    key = "seq" if seq else "y"
    assert gc.maxdiff(got, g[key + "_f64"]) < LOOP_TOL
    assert gc.maxdiff(got, g[key + "_f32"]) < LOOP_TOL


def test_sampler_graph_replay_equals_eager(model, gc, sched):
    g = load_golden("sample_s1")
    rows, _ = gc.table_rows(sched, 4)
    noise = torch.from_numpy(gc.exec_order_noise(gc.noise_from_seed(int(g["seed"]), 2, 6, 4))).cuda()
    args = dict(x_T=torch.from_numpy(g["x_T"]).cuda(), noise=noise)
    with torch.no_grad():
        a = model.sample(torch.from_numpy(g["mel"]).cuda(), rows, **args)
        a2 = model.sample(torch.from_numpy(g["mel"]).cuda(), rows, **args)     # cached graph, second replay
        model.set_option("graph", "0")
        try:
            b = model.sample(torch.from_numpy(g["mel"]).cuda(), rows, **args)
        finally:
            model.set_option("graph", "1")
    assert torch.equal(a, b) and torch.equal(a, a2)


def test_host_checked_fallback_equals_inline_fallback(gc, sched, oracle64):
    """Option fallback = "host": the sampler enqueues no fp32 launch behind its fp16x2 kernels; the range
    flags are read back after the call and a flagged call is redone with those stages on fp32.  With weights that drive the
    activations past the fp16 range the result must be the inline-fallback result (option "graph") up to fp32 rounding, the
    float64 oracle's relative to its scale, and check() must say that it redid the call; with ordinary weights nothing is redone
    and both modes agree bit for bit."""
    import synth
    rows, table = gc.table_rows(sched, 4)
    B, T, N = 2, 7, 4
    mel = synth.synth_mel(31, B, T)
    x_T = synth.hash_normal(31, 1, B * T * 256).reshape(B, 1, T * 256)
    z = gc.noise_from_seed(31, B, T, N)
    args = dict(x_T=torch.from_numpy(x_T).cuda(), noise=torch.from_numpy(gc.exec_order_noise(z)).cuda())
    melc = torch.from_numpy(mel).cuda()
    # ordinary weights: no redo, identical bits
    m = gc.make_model()
    m.set_option("fallback", "host")
    with torch.no_grad():
        a = m.sample(melc, rows, defer_check=True, **args)
        assert m.check() is False
        m.set_option("fallback", "graph")
        b = m.sample(melc, rows, **args)
        m.set_option("fallback", "host")
    assert torch.equal(a, b)
    # a first conv scaled by 3e5: DBlocks, ConvTranspose and the LVC layers of hop 64 / 256 leave the fp16 range
    sd = dict(synth.synth_state_dict(1234))
    sd["first_audio_conv.weight_g"] = (sd["first_audio_conv.weight_g"] * 3.0e5).astype(np.float32)
    m2 = gc.fastdiff_amd.FastDiff()
    m2.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m2 = m2.cuda().eval()
    m2.set_option("fallback", "host")
    o = type(oracle64)("f64")
    o.set_weights(sd)
    ref = o.sample(mel, table, x_T, z)
    with torch.no_grad():
        h1 = m2.sample(melc, rows, defer_check=True, **args)
        assert m2.check() is True                                      # flags were raised: the call ran a second time
        h2 = m2.sample(melc, rows, **args)                             # the same with the check inside sample()
        m2.set_option("fallback", "graph")
        g = m2.sample(melc, rows, **args)
        m2.set_option("fallback", "host")
    scale = float(np.abs(ref).max())
    assert np.isfinite(h1.cpu().numpy()).all() and torch.equal(h1, h2)
    assert gc.maxdiff(h1.cpu().numpy(), ref) < 1e-5 * scale and gc.maxdiff(g.cpu().numpy(), ref) < 1e-5 * scale
    # the pipelined form: the epilogue does not wait for the check (its result is provisional with the waveform); settle(ticket) says
    # whether the call was run again -- here it was -- and the epilogue is then computed again from the final waveform
    with torch.no_grad():
        h3 = m2.sample(melc, rows, defer_check=True, **args)
        ticket = m2.last_ticket
        m2.peak_normalize_int16(h3)
        assert m2.settle(ticket) is True and m2.settle(ticket) is True          # asking twice does no harm
        pcm = m2.peak_normalize_int16(h3)
    assert torch.equal(h3, h1) and pcm.shape == (B, T * 256)
    assert torch.equal(pcm, m2.peak_normalize_int16(h1))


def test_pipelined_host_check_looks_at_a_call_after_the_next_one_is_enqueued(gc, sched):
    """Option fallback = "host" with deferred checks, the way a serving loop drives it: call k is looked at by call k + 1 after that
    one has enqueued itself.  A run of calls that alternate between an ordinary start and one scaled by 1e6 (every DBlock /
    ConvTranspose / LVC launch of that call leaves the fp16 range) must give, call for call, the result of the in-graph fallbacks (the
    same bits where nothing was flagged); settle(ticket) must name exactly the scaled calls, whenever it is asked."""
    import synth
    rows, _ = gc.table_rows(sched, 4)
    B, T, N = 2, 37, 4
    mel = torch.from_numpy(synth.synth_mel(41, B, T)).cuda()
    starts = []
    for k in range(6):
        x = synth.hash_normal(50 + k, 1, B * T * 256).reshape(B, 1, T * 256)
        starts.append(torch.from_numpy(x * (1.0e6 if k % 2 else 1.0)).float().cuda())
    zeros = torch.zeros(N, B, 1, T * 256, device="cuda")
    ref_model = gc.make_model()
    ref_model.set_option("fallback", "graph")
    with torch.no_grad():
        want = [ref_model.sample(mel, rows, x_T=x, noise=zeros) for x in starts]
        flags = ref_model.read_tap("range_flags_call").view(np.int32)
    assert flags[13:19].all()                                   # the last (scaled) call did leave the range
    m = gc.make_model()
    m.set_option("fallback", "host")
    got, tickets = [], []
    with torch.no_grad():
        for x in starts:
            got.append(m.sample(mel, rows, x_T=x, noise=zeros, defer_check=True))
            tickets.append(m.last_ticket)
        redone = [m.settle(t) for t in tickets]                 # the last call is looked at here, the others already were
    assert redone == [False, True, False, True, False, True]
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(got, want)):
        if k % 2 == 0:
            assert torch.equal(a, b), k                          # nothing flagged: the same kernels ran
        else:                                                    # redone with the flagged stages on fp32 in EVERY step (in-graph: per step)
            scale = float(b.abs().max())
            assert torch.isfinite(a).all() and float((a - b).abs().max()) < 1e-5 * scale, (k, scale)
    assert tickets == list(range(tickets[0], tickets[0] + 6)) and m.settle(tickets[1]) is True


def test_host_checked_fallback_long_schedule(gc, sched):
    """N = 20 > 8: under fallback = host the schedule runs as 8-step pieces, each checked before the next starts; an overflow that
    appears in the middle (a table whose c_div = 0.4 lets x grow 2.5x per step: past the fp16 range around step 10) costs one
    repeated piece, the rest runs with the flagged kernels on fp32.  Against the inline-fallback run of the same call."""
    import synth
    m = gc.make_model()
    m.set_option("fallback", "host")
    B, T, N = 1, 6, 20
    mel = torch.from_numpy(synth.synth_mel(41, B, T)).cuda()
    x_T = torch.from_numpy(synth.hash_normal(41, 1, B * T * 256).reshape(B, 1, T * 256)).cuda()
    rows = [{"t": 400.0 - 10 * k, "c_eps": 0.01, "c_div": 0.4, "sigma": 0.1, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": int(k < N - 1)} for k in range(N)]
    noise = torch.from_numpy(np.stack([synth.hash_normal(41, 2 + k, B * T * 256).reshape(B, 1, T * 256) for k in range(N)])).cuda()
    with torch.no_grad():
        a = m.sample(mel, rows, x_T=x_T, noise=noise, return_sequence=True)
        m.set_option("fallback", "graph")
        b = m.sample(mel, rows, x_T=x_T, noise=noise, return_sequence=True)
        flags = m.read_tap("range_flags_call").view(np.int32)
    peak = [float(v.abs().max()) for v in b]
    assert peak[-1] > 1.0e6 and peak[4] < 3.0e3, peak[::4]             # the range is left somewhere in the middle
    assert flags[13:19].all()                                          # ... and the inline run did fall back at the end
    for k in range(N + 1):
        assert torch.isfinite(a[k]).all()
        assert float((a[k] - b[k]).abs().max()) <= 1e-4 * max(1.0, peak[k]), k
    assert torch.equal(a[1], b[1])                                     # before anything overflows the two modes run the same kernels


@pytest.mark.parametrize("N", [4, 19])
def test_hoisted_predictor_equals_the_per_step_one(gc, sched, N):
    """The predictor never sees x, so fd_sample runs it for all reverse steps of a short schedule at once (N <= 8), and for a longer
    one at the front of every captured 8-step piece (N = 19: pieces of 8, 8 and 3 steps, rows step_idx .. of the embedding table read
    through the device step counter).  Same kernels on the same operands as the per-step predictor (option hoist = off): the
    same bits, on a ragged batch, from the graph and launched one by one."""
    import synth
    B, T = 3, 37
    lens = [37, 12, 25]
    mel = torch.from_numpy(synth.synth_mel(77, B, T)).cuda()
    for b, t in enumerate(lens):
        mel[b, :, t:] = 0.0
    rows = [{"t": 190.0 - 9.5 * k, "c_eps": 0.02, "c_div": 0.99, "sigma": 0.05, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": int(k < N - 1)}
            for k in range(N)]
    out = {}
    with torch.no_grad():
        # fallback = graph with the launches one by one runs ALL N steps as one sequence (round-3 ADVICE: the deferred bookkeeping must
        # not cross a piece boundary there, or the second piece's predictor reads embedding rows shifted by one step)
        for fallback in ("host", "graph"):
            for hoist in ("on", "off"):
                for graph in ("1", "0"):
                    m = gc.make_model()
                    m.set_option("fallback", fallback)
                    m.set_option("hoist", hoist)
                    m.set_option("graph", graph)
                    y = m.sample(mel, rows, seed=5, lens=lens)
                    y2 = m.sample(mel, rows, seed=5, lens=lens)          # (second call: cached graphs, workspace already sized)
                    assert torch.equal(y, y2), (fallback, hoist, graph)
                    out[(fallback, hoist, graph)] = y
    ref = out[("host", "off", "0")]
    assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0.1
    for k, y in out.items():
        assert torch.equal(y, ref), k


@pytest.mark.parametrize("B,T,lens", [(3, 37, [37, 12, 25]), (2, 130, None), (1, 5, None), (4, 64, [64, 1, 33, 64]), (3, 230, [112, 230, 113])])
def test_up_sampler_inside_the_first_lvc_layer(gc, sched, B, T, lens):
    """Under the host-checked range (every stage fp16x2-only) blocks 1 and 2 run their ConvTranspose inside the first LVC layer
    (k_lvc_h2<.., UP>): the same matrix instructions on the same operands as k_convt_h2, so option fuse_up = 0 must give the same
    bits -- tile edges, ragged lengths and one-frame utterances included."""
    import synth
    mel = torch.from_numpy(synth.synth_mel(500 + T, B, T)).cuda()
    if lens:
        for b, t in enumerate(lens):
            mel[b, :, t:] = 0.0
    rows, _ = gc.table_rows(sched, 4)
    out = {}
    with torch.no_grad():
        for fuse in ("1", "0"):
            m = gc.make_model()
            assert m._options.get("fallback") == "host"
            m.set_option("fuse_up", fuse)
            out[fuse] = [m.sample(mel, rows, seed=11, lens=lens), m.sample(mel, rows, seed=12, lens=lens)]
    for a, b in zip(out["1"], out["0"]):
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0.1
        assert torch.equal(a, b)


@pytest.mark.parametrize("N", [4, 19])
def test_step_bookkeeping_in_the_next_steps_first_kernel(gc, N):
    """Between two steps of one graph (or launch sequence) the end-of-step bookkeeping -- next row of the step table, rotate the range
    flags -- rides in the next step's first kernel; only the last step of a sequence keeps the k_advance launch.  Option
    fuse_advance = 0 (a launch after every step) must give the same trajectory, from the graph and launched one by one."""
    import synth
    B, T = 2, 9
    mel = torch.from_numpy(synth.synth_mel(88, B, T)).cuda()
    rows = [{"t": 190.0 - 9.5 * k, "c_eps": 0.02, "c_div": 0.99, "sigma": 0.05, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": int(k < N - 1)}
            for k in range(N)]
    out = {}
    with torch.no_grad():
        for adv in ("1", "0"):
            for graph in ("1", "0"):
                m = gc.make_model()
                m.set_option("fuse_advance", adv)
                m.set_option("graph", graph)
                out[(adv, graph)] = torch.stack(list(m.sample(mel, rows, seed=9, return_sequence=True)))
    ref = out[("0", "0")]
    assert torch.isfinite(ref).all() and float((ref[-1] - ref[0]).abs().max()) > 0.1
    for k, y in out.items():
        assert torch.equal(y, ref), k


def test_embedding_table_kept_between_calls(gc, sched):
    """The step-embedding rows of a schedule (two launches per call) stay in the workspace while the next call uses the same t values
    and batch size; a different schedule, another batch size or an fd_forward in between must bring the table up to date."""
    import synth
    B, T = 2, 21
    mel = torch.from_numpy(synth.synth_mel(61, B, T)).cuda()
    rows4, _ = gc.table_rows(sched, 4)
    rows6, _ = gc.table_rows(sched, 6)
    rows19 = [{"t": 190.0 - 9.5 * k, "c_eps": 0.02, "c_div": 0.99, "sigma": 0.05, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": int(k < 18)}
              for k in range(19)]
    with torch.no_grad():
        want = {n: gc.make_model().sample(mel, r, seed=3) for n, r in (("4", rows4), ("6", rows6), ("19", rows19))}      # fresh handles
        want1 = gc.make_model().sample(mel[:1], rows4, seed=3)
        m = gc.make_model()
        for n, r in (("4", rows4), ("4", rows4), ("6", rows6), ("4", rows4), ("19", rows19), ("19", rows19), ("6", rows6)):
            assert torch.equal(m.sample(mel, r, seed=3), want[n]), n
        assert torch.equal(m.sample(mel[:1], rows4, seed=3), want1)              # another batch size
        audio = torch.from_numpy(synth.synth_audio(61, B, T)).cuda()
        m((audio, mel, torch.full((B, 1), 7.0).cuda()))                           # fd_forward writes its own rows into the table
        assert torch.equal(m.sample(mel, rows4, seed=3), want["4"])


def test_frame_bucketed_call_equals_the_exact_length_call(gc, sched):
    """The reference CLI vocodes one utterance per call, a different length each time (FastDiff.py:97-103, base.yaml:53,
    dataset_utils.py:114-125).  fd_sample sizes its own buffers and graphs for T rounded up to a multiple of 32 frames (option
    t_bucket) and runs the call with `lens`; the caller's tensors keep their dense layout.  For 20 random lengths the bucketed call
    must be torch.equal to the exact-T call (t_bucket = 0) -- with device noise (flat and per-utterance streams), with injected x_T / z
    and the returned sequence, for a batch of two with and without `lens` -- and one graph must have served every length of a bucket."""
    import synth
    rng = np.random.default_rng(20)
    lengths = sorted(set([3, 31, 32, 33, 64, 95] + rng.integers(3, 200, 20).tolist()))
    rows, _ = gc.table_rows(sched, 4)
    rows6, _ = gc.table_rows(sched, 6)
    exact, bucketed = gc.make_model(), gc.make_model()
    exact.set_option("t_bucket", "0")
    with torch.no_grad():
        for m in (exact, bucketed):      # the workspace at its final size first: growing it drops the graphs captured so far
            m.sample(torch.zeros(2, 80, max(lengths)).cuda(), rows6, seed=1)
        base = {m: m.counter("graph_captures") for m in (exact, bucketed)}
        for T in lengths:
            mel = torch.from_numpy(synth.synth_mel(300 + T, 2, T)).cuda()
            L = T * 256
            x_T = torch.from_numpy(synth.hash_normal(T, 1, 2 * L).reshape(2, 1, L)).cuda()
            z = torch.from_numpy(np.stack([synth.hash_normal(T, 2 + k, 2 * L).reshape(2, 1, L) for k in range(4)])).cuda()
            lens = [T, max(1, T - 1 - T // 3)]
            cases = {
                "philox_b1": lambda m: m.sample(mel[:1], rows, seed=5),
                "philox_b2_streams": lambda m: m.sample(mel, rows, seed=6, stream_ids=[11, 4]),
                "injected_seq": lambda m: torch.stack(list(m.sample(mel, rows, x_T=x_T, noise=z, return_sequence=True))),
                "ragged_lens": lambda m: m.sample(mel, rows6, seed=7, lens=lens, stream_ids=[0, 1]),
                "ddim": lambda m: m.sample(mel[:1], rows, seed=8, ddim=True),
            }
            for name, fn in cases.items():
                a, b = fn(exact), fn(bucketed)
                if name == "ragged_lens":      # behind an utterance's own length the output is unspecified
                    a, b = [a[i, :, : lens[i] * 256] for i in range(2)], [b[i, :, : lens[i] * 256] for i in range(2)]
                    assert all(torch.equal(x, y) for x, y in zip(a, b)), (T, name)
                else:
                    assert torch.isfinite(a).all() and torch.equal(a, b), (T, name)
    buckets = len({(T + 31) // 32 for T in lengths})
    # per bucket: (B=1, N=4) shared by the DDPM and "ddim" calls, (B=2, N=4) with and without streams, (B=2, N=6): three graphs (a
    # length that IS a multiple of 32 runs without `lens` and has its own); the exact-T handle captured three per LENGTH
    captures = (bucketed.counter("graph_captures"), exact.counter("graph_captures"))
    with torch.no_grad():
        # a long schedule (N = 19: two 8-step pieces + a remainder, each with its own hoisted predictor) and the benchmark's own length - 1
        rows19 = [{"t": 190.0 - 9.5 * k, "c_eps": 0.02, "c_div": 0.99, "sigma": 0.05, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": int(k < 18)}
                  for k in range(19)]
        mel45 = torch.from_numpy(synth.synth_mel(45, 2, 45)).cuda()
        a, b = exact.sample(mel45, rows19, seed=3, lens=[45, 17]), bucketed.sample(mel45, rows19, seed=3, lens=[45, 17])
        assert torch.equal(a[0], b[0]) and torch.equal(a[1, :, : 17 * 256], b[1, :, : 17 * 256])
        mel863 = torch.from_numpy(synth.synth_mel(863, 1, 863)).cuda()
        a, b = exact.sample(mel863, rows, seed=9), bucketed.sample(mel863, rows, seed=9)
        assert torch.isfinite(a).all() and torch.equal(a, b)
    exact_mult = sum(1 for T in lengths if T % 32 == 0)
    got_b, got_e = captures[0] - base[bucketed], captures[1] - base[exact]
    assert 3 * buckets - 1 <= got_b <= 3 * buckets + 2 * exact_mult, (got_b, buckets)      # (- 1: the warm-up call's own graph)
    assert 3 * len(lengths) - 1 <= got_e <= 3 * len(lengths), (got_e, len(lengths))


def test_graph_cache_evicts_without_waiting_and_keeps_results(gc, sched):
    """A graph cache of 2 under a stream of calls from 5 buckets: every call's result equals a fresh handle's, evicted graphs are
    retired behind an event (not destroyed under a running replay, no stream-wide wait) and reaped by later calls."""
    import synth
    rows, _ = gc.table_rows(sched, 4)
    lengths = [20, 40, 70, 100, 130]
    mels = {T: torch.from_numpy(synth.synth_mel(T, 1, T)).cuda() for T in lengths}
    with torch.no_grad():
        want = {T: gc.make_model().sample(mels[T], rows, seed=T) for T in lengths}
        m = gc.make_model()
        m.set_option("graph_cache", "2")
        m.sample(mels[130], rows, seed=1)      # the workspace at its final size (growing it drops every graph, uncounted)
        outs = []
        for rnd in range(4):
            for T in lengths:
                outs.append((T, m.sample(mels[T], rows, seed=T, defer_check=True)))
        m.check()
        torch.cuda.synchronize()
    for T, y in outs:
        assert torch.equal(y, want[T]), T
    assert m.counter("graphs_resident") == 2
    # 5 buckets in a cycle through an LRU of 2: every call misses (the warm-up's graph is gone by the time its bucket comes round)
    assert m.counter("graph_captures") == 21 and m.counter("graph_evictions") == 19 and m.counter("graph_hits") == 0
    with torch.no_grad():
        m.sample(mels[20], rows, seed=20)      # everything retired so far has completed: the next look-up that misses reaps it
        torch.cuda.synchronize()
        m.sample(mels[70], rows, seed=70)
    assert m.counter("graphs_retired") <= 2


def test_consecutive_calls_may_change_streams(gc, sched):
    """One handle, calls on different streams (fd_api.cpp: follow_stream): a call arriving on another stream than the previous one settles
    the pending range check and waits for the tail of the previous call (an event recorded at the end of every call) -- deferred checks,
    cached graphs, the shared workspace and the embedding table included.  Every result must equal a fresh handle's."""
    import synth
    B, T = 2, 33
    mel = torch.from_numpy(synth.synth_mel(77, B, T)).cuda()
    mel2 = torch.from_numpy(synth.synth_mel(78, 1, 50)).cuda()
    rows, _ = gc.table_rows(sched, 4)
    rows6, _ = gc.table_rows(sched, 6)
    audio = torch.from_numpy(synth.synth_audio(77, B, T)).cuda()
    steps = torch.full((B, 1), 7.0).cuda()
    with torch.no_grad():
        want_a = gc.make_model().sample(mel, rows, seed=3)
        want_b = gc.make_model().sample(mel2, rows6, seed=4)
        want_f = gc.make_model()((audio, mel, steps))
        m = gc.make_model()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        got = []
        for rnd in range(3):
            with torch.cuda.stream(s1):
                a = m.sample(mel, rows, seed=3, defer_check=True)          # its check is still pending when the next call arrives on s2
            with torch.cuda.stream(s2):
                b = m.sample(mel2, rows6, seed=4, defer_check=(rnd == 1))
            f = m((audio, mel, steps))                                     # default stream
            with torch.cuda.stream(s1):
                a2 = m.sample(mel, rows, seed=3)
            got.append((a, b, f, a2))
        torch.cuda.synchronize()
        assert m.check() is False
        for a, b, f, a2 in got:
            assert torch.equal(a, want_a) and torch.equal(a2, want_a) and torch.equal(b, want_b) and torch.equal(f, want_f)


def test_a_stream_destroyed_between_two_calls():
    """The library must not touch the stream of an earlier, settled call: the caller may have destroyed it (a hipEventRecord on a destroyed
    stream handle takes the process down on this runtime -- which is why this runs in a process of its own)."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_switch_probe.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "destroyed stream: ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_graph_cache_alternating_shapes(gc, sched):
    """One captured step per (B, T, mode) is kept (micro-batches of different padded length alternate in infer.py): results with
    the cache warm, after other shapes ran in between, and after more shapes than the cache holds (eviction) must equal the
    first, freshly captured run bit for bit."""
    import synth
    m = gc.make_model()
    rows, _ = gc.table_rows(sched, 4)
    shapes = [(2, 5), (1, 9), (3, 2)] + [(1, t) for t in range(10, 30)]          # 23 shapes > 16 cache entries
    first = {}
    with torch.no_grad():
        for rnd in range(2):
            for (B, T) in shapes if rnd == 0 else shapes[:3] + shapes[-2:] + shapes[:3]:
                mel = torch.from_numpy(synth.synth_mel(300 + 10 * B + T, B, T)).cuda()
                y = m.sample(mel, rows, seed=B * 100 + T)
                if (B, T) in first:
                    assert torch.equal(y, first[(B, T)]), (rnd, B, T)
                else:
                    first[(B, T)] = y.clone()
    assert all(torch.isfinite(v).all() for v in first.values())


def test_n1000_full_schedule_drift(model, gc, sched):
    """BASELINE config 3 (long-loop hipGraph stress): N=1000, drift bounded by 10x the fp32 reference's own."""
    g = load_golden("sample_s4")
    rows, _ = gc.table_rows(sched, 1000)
    noise = gc.exec_order_noise(gc.noise_from_seed(int(g["seed"]), 1, 4, 1000))
    with torch.no_grad():
        y = model.sample(torch.from_numpy(g["mel"]).cuda(), rows, x_T=torch.from_numpy(g["x_T"]).cuda(),
                         noise=torch.from_numpy(noise).cuda())
    y = y.cpu().numpy()
    ref_drift = gc.maxdiff(g["y_f32"], g["y_f64"])
    assert np.isfinite(y).all()
    assert gc.maxdiff(y, g["y_f64"]) < 10 * ref_drift, (gc.maxdiff(y, g["y_f64"]), ref_drift)


def test_n1000_at_64_frames_against_the_reference_trajectory(gc, sched):
    """BASELINE configs[2] with tile edges: 64 frames (16 tiles per row at hop 256, partial workgroups at hop 8), the full N = 1000
    schedule, against the reference's own float64 trajectory every 125 steps (tests/golden/sample_s6.npz).  Bar: 10x the float32
    reference's own distance from its float64 run at the same checkpoint (floor 2e-5).  The same call with every contraction on
    the exact-fp32 pipe must meet the same bar; the range flags say which launches, if any, handed over to their fp32 kernels."""
    g = load_golden("sample_s6")
    N = int(g["N"])
    B, _, T = g["mel"].shape
    rows, _ = gc.table_rows(sched, N)
    noise = torch.from_numpy(gc.exec_order_noise(gc.noise_from_seed(int(g["seed"]), B, T, N))).cuda()
    mel, x_T = torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["x_T"]).cuda()
    idx = [int(k) for k in g["ckpt_idx"]]
    worst = {}
    for pipe in ("f16x2", "fp32"):
        m = gc.make_model()
        for k in ("gemm", "lvc", "conv"):
            m.set_option(k, pipe)
        with torch.no_grad():
            seq = m.sample(mel, rows, x_T=x_T, noise=noise, return_sequence=True)
        flags = m.read_tap("range_flags_call").view(np.int32)
        for i, k in enumerate(idx):
            got = seq[k].cpu().numpy()
            assert np.isfinite(got).all()
            ref_drift = gc.maxdiff(g["ckpt_f32"][i], g["ckpt_f64"][i])
            d = gc.maxdiff(got, g["ckpt_f64"][i])
            assert d <= max(10 * ref_drift, 2e-5), (pipe, k, d, ref_drift)
            worst[(pipe, k)] = (d, ref_drift)
        if pipe == "fp32":
            assert not flags.any()            # the fp32 kernels have no operand range to watch
        else:
            # this trajectory stays small (|x| <= 631 at x_0): no launch may have left the fp16 pipe
            assert not flags.any(), np.nonzero(flags)[0]
    print("n1000 T=64 worst (ours vs f64, reference f32 vs f64):", {k: (f"{v[0]:.2e}", f"{v[1]:.2e}") for k, v in worst.items() if k[1] in (125, 1000)})


def test_config3_n1000_at_864_frames_against_the_reference_trajectory(gc, sched):
    """BASELINE configs[2] END TO END at its own size (round-5 VERDICT item 3): B = 1, T = 864 (221,184 samples), all 1000 reverse steps
    of linspace(1e-6, 0.01, 1000) (FastDiff.py:76-78; util.py:158-235) on the default pipe, against the trajectory the REFERENCE itself
    produced in float64 on the same weights, mel and injected noise (tests/golden/sample_s7.npz, oracle/gen_golden.py sample_long).
    Weights: the contractive synthetic set (synth.make_contractive) -- eps is positively correlated with x as a trained denoiser's is, so
    |x| stays O(1) for the whole schedule instead of growing to hundreds, and nothing may leave the fp16x2 pipe.
    Bar at every stored state (every 125 steps, every 7th sample; x_0 in full): 10x the float32 reference's own distance from its
    float64 run there (floor 2e-5).  The injected noise (885 MB) is rebuilt on the device with the torch twin of the synthetic hash."""
    g = load_golden("sample_s7")
    N, seed, sub = int(g["N"]), int(g["seed"]), int(g["sub"])
    B, _, T = g["mel"].shape
    assert (B, T, N) == (1, 864, 1000)
    L = T * 256
    rows, _ = gc.table_rows(sched, N)
    x_T = gc.hash_normal_torch(seed, 1, L).view(1, 1, L)
    noise = torch.zeros((N, 1, 1, L), dtype=torch.float32, device="cuda")      # noise[k]: added after executed step k = reverse index N-1-k
    for k in range(N - 1):
        noise[k, 0, 0] = gc.hash_normal_torch(seed, 2 + (N - 1 - k), L)
    m = gc.make_model(contractive=True)
    with torch.no_grad():
        seq = m.sample(torch.from_numpy(g["mel"]).cuda(), rows, x_T=x_T, noise=noise, return_sequence=True)
    info = {"pieces": m.counter("pieces"), "pieces_redone": m.counter("pieces_redone"), "pieces_fp32": m.counter("pieces_fp32")}
    flags = m.read_tap("range_flags_call").view(np.int32)
    report = {}
    for i, k in enumerate(int(v) for v in g["ckpt_idx"]):
        got = seq[k].cpu().numpy()
        assert np.isfinite(got).all(), k
        bar = max(10 * float(g["ckpt_ref_drift"][i]), 2e-5)
        d = gc.maxdiff(got[..., ::sub], g["ckpt_f64_sub"][i])
        assert d <= bar + 1e-7 * float(g["ckpt_peak"][i]), (k, d, bar)      # (the stored states are float32 roundings of the float64 run)
        report[k] = (d, float(g["ckpt_ref_drift"][i]), float(g["ckpt_peak"][i]))
    d0 = gc.maxdiff(seq[N].cpu().numpy(), g["y_f64"])
    ref0 = gc.maxdiff(g["y_f32"], g["y_f64"])
    print("config3 T=864 N=1000 vs the reference's float64 run: step -> (ours, reference f32, max|x|):",
          {k: (f"{v[0]:.2e}", f"{v[1]:.2e}", f"{v[2]:.3g}") for k, v in report.items()}, f"; x_0 in full: ours {d0:.3e}, reference f32 {ref0:.3e};", info)
    assert d0 <= max(10 * ref0, 2e-5), (d0, ref0)
    assert float(np.abs(g["y_f64"]).max()) <= 2.0                           # the trajectory contracted: a waveform-sized x_0
    assert info == {"pieces": 125, "pieces_redone": 0, "pieces_fp32": 0} and not flags.any(), (info, np.nonzero(flags)[0])


def test_config3_n1000_at_864_frames_both_pipes_and_first_steps_against_oracle(gc, sched, oracle64):
    """BASELINE configs[2] at its own size: B = 1, T = 864 (221,184 samples), the full N = 1000 schedule (FastDiff.py:76-78;
    util.py:158-235), round-3 VERDICT item 1a.
    (1) SAME-PRODUCT CROSS-PIPE CHECK: the default pipe (fp16x2 contractions, host-checked hand-over to fp32) against every
        contraction on the exact-fp32 kernels, same device noise (Philox streams are keyed on seed / step / sample, not on the pipe).
        Bar: the distance relative to max|x_0| within 10x the reference's own float32-vs-float64 relative drift after 1000 steps
        (tests/golden/sample_s6.npz: 8.2e-4 on |x_0| <= 631 = 1.3e-6).  Reads the call's range flags and how many of its 125
        eight-step pieces were run again / ran with stages already on fp32 (fd_get_counter) and prints them.
    (2) ORACLE: the first 16 reverse steps of that schedule (t = 999 .. 984) with injected x_T and z against the float64 C oracle,
        every step of the sequence (the 16th without its noise draw, on both sides)."""
    import synth
    g6 = load_golden("sample_s6")
    ref_rel = gc.maxdiff(g6["ckpt_f32"][-1], g6["ckpt_f64"][-1]) / float(np.abs(g6["ckpt_f64"][-1]).max())
    B, T, N = 1, 864, 1000
    rows, table = gc.table_rows(sched, N)
    mel_np = synth.synth_mel(33, B, T)
    mel = torch.from_numpy(mel_np).cuda()
    ys, info = {}, {}
    for pipe in ("f16x2", "fp32"):
        m = gc.make_model()
        for k in ("gemm", "lvc", "conv"):
            m.set_option(k, pipe)
        with torch.no_grad():
            y = m.sample(mel, rows, seed=31, stream_ids=[0])
        flags = m.read_tap("range_flags_call").view(np.int32)
        info[pipe] = {"flag_words": np.nonzero(flags)[0].tolist(), "pieces": m.counter("pieces"), "pieces_redone": m.counter("pieces_redone"),
                      "pieces_fp32": m.counter("pieces_fp32"), "fp32_mask": hex(m.counter("fp32_mask"))}
        ys[pipe] = y.cpu().numpy().astype(np.float64)
        assert np.isfinite(ys[pipe]).all(), pipe
        del m
    peak = float(np.abs(ys["fp32"]).max())
    d = float(np.abs(ys["f16x2"] - ys["fp32"]).max())
    print(f"config3 T=864 N=1000: cross-pipe max|d| = {d:.3e} on max|x_0| = {peak:.4g} (relative {d / peak:.2e}; reference fp32-vs-fp64 "
          f"relative drift at 1000 steps {ref_rel:.2e}); default pipe {info['f16x2']}; fp32 pipe {info['fp32']}")
    assert info["f16x2"]["pieces"] == 125 and info["fp32"]["pieces"] == 125
    assert info["fp32"]["pieces_redone"] == 0 and not info["fp32"]["flag_words"]
    assert info["f16x2"]["pieces_redone"] + info["f16x2"]["pieces_fp32"] <= 125
    assert d / peak <= 10 * ref_rel, (d, peak, ref_rel)
    # (2) the first 16 steps against the float64 oracle
    K = 16
    top = {k: np.asarray(v)[N - K:] for k, v in table.items()}            # reverse indices N-16 .. N-1, executed last to first
    rows16 = [dict(r) for r in rows[:K]]
    rows16[-1]["add_noise"] = 0
    x_T = synth.hash_normal(34, 1, B * T * 256).reshape(B, 1, T * 256)
    z = gc.noise_from_seed(34, B, T, K)                                   # z[n] added after reverse index n > 0 of the 16-step table
    ref = oracle64.sample(mel_np, top, x_T, z, return_sequence=True)
    m = gc.make_model()
    with torch.no_grad():
        seq = m.sample(mel, rows16, x_T=torch.from_numpy(x_T).cuda(), noise=torch.from_numpy(gc.exec_order_noise(z)).cuda(), return_sequence=True)
    worst = 0.0
    for k in range(K + 1):
        dk = gc.maxdiff(seq[k].cpu().numpy(), ref[k])
        worst = max(worst, dk)
        assert dk <= LOOP_TOL * max(1.0, float(np.abs(ref[k]).max()) / 8.0), (k, dk)
    print(f"config3 T=864: first 16 steps vs float64 oracle, worst max|d| over the sequence = {worst:.3e} (max|x| {np.abs(ref[-1]).max():.3f})")
    assert not m.read_tap("range_flags_call").view(np.int32).any()


def test_drop_in_sampling_function(model, gc, sched, oracle64):
    """The reference call shape: sampling_given_noise_schedule(net, size, dh, schedule, condition=mel)."""
    import fastdiff_amd
    import synth
    B, T = 2, 5
    mel = synth.synth_mel(21, B, T)
    dh = fastdiff_amd.schedules.training_hyperparams()
    sch_t = fastdiff_amd.schedules.noise_schedule_for(4).cuda()
    x_T = synth.hash_normal(5, 1, B * T * 256).reshape(B, 1, T * 256)
    z = gc.noise_from_seed(5, B, T, 4)
    y = fastdiff_amd.sampling_given_noise_schedule(model, (B, 1, T * 256), dh, sch_t, condition=torch.from_numpy(mel).cuda(),
                                                   x_T=torch.from_numpy(x_T).cuda(),
                                                   noise=torch.from_numpy(gc.exec_order_noise(z)).cuda(), verbose=False)
    _, table = gc.table_rows(sched, 4)
    ref = oracle64.sample(mel, table, x_T, z)
    assert gc.maxdiff(y.cpu().numpy(), ref) < LOOP_TOL
    # on-device Philox: reproducible per seed, different across seeds, finite
    kw = dict(condition=torch.from_numpy(mel).cuda(), verbose=False)
    a = fastdiff_amd.sampling_given_noise_schedule(model, (B, 1, T * 256), dh, sch_t, seed=3, **kw)
    b = fastdiff_amd.sampling_given_noise_schedule(model, (B, 1, T * 256), dh, sch_t, seed=3, **kw)
    c = fastdiff_amd.sampling_given_noise_schedule(model, (B, 1, T * 256), dh, sch_t, seed=4, **kw)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(a).all()
    # the reference's own random stream (CPU generator, reference order)
    torch.manual_seed(99)
    r1 = fastdiff_amd.sampling_given_noise_schedule(model, (B, 1, T * 256), dh, sch_t, noise_source="reference", **kw)
    torch.manual_seed(99)
    xT = torch.normal(0, 1, size=(B, 1, T * 256))
    zs = [torch.normal(0, 1, size=(B, 1, T * 256)) for _ in range(3)] + [torch.zeros(B, 1, T * 256)]
    r2 = fastdiff_amd.sampling_given_noise_schedule(model, (B, 1, T * 256), dh, sch_t, x_T=xT.cuda(),
                                                    noise=torch.stack(zs).cuda(), **kw)
    assert torch.equal(r1, r2)


@pytest.mark.parametrize("case,hp", [("N4", {"N": 4, "noise_schedule": ""}), ("N6", {"N": 6, "noise_schedule": ""}),
                                     ("list3", {"N": 4, "noise_schedule": [9.0000e-05, 9.0000e-03, 6.0000e-01]})])
def test_caller_test_step_against_the_references_own(model, gc, monkeypatch, tmp_path, case, hp):
    """The caller of the hot path, pinned on the reference's own function: tests/golden/test_step.npz holds what
    FastDiffTask.test_step (modules/FastDiff/task/FastDiff.py:60-119, cut out with ast and executed on the reference model) wrote
    through save_wav -- schedule by hparams['N'] or the hparams['noise_schedule'] list, size (1, 1, T * hop_size), peak-normalised
    int16.  infer.test_step mirrors it around the HIP sampler; with std_normal replaying the same draws in the reference's order
    (noise_source = "reference") the file it writes must hold the same samples within 1 LSB."""
    import synth
    from scipy.io import wavfile
    from fastdiff_amd import infer, sampler
    g = load_golden("test_step")
    mel, seed = g["mel"], int(g["seed"])
    L = mel.shape[-1] * 256
    n_draws = int(g["n_draws_" + case])
    draws = iter([synth.hash_normal(seed, 1, L)] + [synth.hash_normal(seed, 2 + n, L) for n in range(n_draws - 1, 0, -1)])
    monkeypatch.setattr(sampler, "std_normal", lambda size: torch.from_numpy(next(draws).copy()).view(*size).cuda())
    hparams = dict(hp, hop_size=256, audio_sample_rate=22050)
    name = str(g["item_name"][0])
    out = infer.test_step(model, {"mels": torch.from_numpy(mel), "wavs": [], "item_name": [name]}, hparams, gen_dir=str(tmp_path), noise_source="reference")
    assert next(draws, None) is None                              # exactly the reference's number of draws
    sr, pcm = wavfile.read(str(tmp_path / f"{name}_pred.wav"))
    assert sr == 22050 and np.array_equal(pcm, out[name])
    d = np.abs(pcm.astype(np.int32) - g["pcm_" + case].astype(np.int32))
    print(f"test_step {case}: int16 max|d| = {int(d.max())}, differing samples {int((d > 0).sum())} of {d.size}")
    assert d.max() <= 1
    with pytest.raises(NotImplementedError):                      # FastDiff.py:92-93
        infer.test_step(model, {"mels": torch.from_numpy(mel), "wavs": [], "item_name": [name]}, dict(hparams, N=5, noise_schedule=""))


def test_noise_scheduling_on_the_hip_denoiser(model, gc, sched, monkeypatch):
    """noise_scheduling (util.py:237-288) with the HIP module as `net` and a stand-in `noise_pred` attached to it: the schedule
    found must be the one the reference function finds on the reference module (golden: gen_noise_scheduling)."""
    import fastdiff_amd
    import synth
    from fastdiff_amd import sampler
    g = load_golden("noise_scheduling")
    monkeypatch.setattr(sampler, "std_normal", lambda size: torch.from_numpy(g["x_T"].copy()).view(*size).cuda())
    dh = {"N": int(g["N"]), "betaN": float(g["betaN"]), "alphaN": float(g["alphaN"]), "rho": float(g["rho"]),
          "alpha": torch.from_numpy(sched["train_alpha"])}
    size = (1, 1, g["x_T"].shape[-1])
    with pytest.raises(AttributeError):                        # the reference's FastDiff has no noise_pred either
        fastdiff_amd.noise_scheduling(model, size, dh, condition=torch.from_numpy(g["mel"]).cuda())
    model.noise_pred = synth.stub_noise_pred
    try:
        for ddim in (False, True):
            betas = fastdiff_amd.noise_scheduling(model, size, dh, condition=torch.from_numpy(g["mel"]).cuda(), ddim=ddim)
            key = "betas_ddim" if ddim else "betas_ddpm"
            assert betas.is_cuda and betas.shape == g[key + "_f32"].shape
            d64 = np.abs(betas.double().cpu().numpy() - g[key + "_f64"]).max()
            ref = np.abs(g[key + "_f32"] - g[key + "_f64"]).max()
            print(key, "max |d beta| vs f64 reference %.2e (fp32 reference: %.2e)" % (d64, ref))
            assert d64 < 1e-5 * g[key + "_f64"].max()
    finally:
        del model.noise_pred


def test_validation_loss_on_the_hip_denoiser(model, gc, sched, monkeypatch):
    """FastDiffTask.validation_step = theta_timestep_loss(model, (mels, wavs), dh) (FastDiff.py:52-57), forward only: with the HIP
    module as `net` the loss and the x_0 estimate must be the reference's (golden: gen_theta_loss, steps and z replayed).  With
    autograd recording the same call builds the graph of fastdiff_amd/train.py; the sampling entry point refuses such inputs."""
    import fastdiff_amd
    from fastdiff_amd import sampler
    g = load_golden("theta_loss")
    monkeypatch.setattr(sampler, "std_normal", lambda size: torch.from_numpy(g["z"].copy()).view(*size).cuda())
    monkeypatch.setattr(torch, "randint", lambda *a, **k: torch.from_numpy(g["ts"].copy()))
    dh = {"T": 1000, "alpha": torch.from_numpy(sched["train_alpha"]).cuda()}
    X = (torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["audio"]).cuda())
    with torch.no_grad():
        loss, x0 = fastdiff_amd.theta_timestep_loss(model, X, dh, reverse=True)
    ref_gap = abs(float(g["loss_f32"]) - float(g["loss_f64"]))
    print("loss %.9f, |d| vs f64 reference %.2e (fp32 reference: %.2e)" % (loss.item(), abs(loss.item() - float(g["loss_f64"])), ref_gap))
    assert abs(loss.item() - float(g["loss_f64"])) < 1e-6 * float(g["loss_f64"])
    # x_0 = (x_t - delta eps) / alpha_t: at t = 999 alpha_t is 0.08, so eps errors are amplified 12x
    assert gc.maxdiff(x0.cpu().numpy(), g["x0_f64"]) < 12.5 * FWD_TOL
    # an input that requires a gradient: the autograd path (fastdiff_amd/train.py, tests/test_training_path.py) -- same loss
    audio = X[1].clone().requires_grad_(True)
    loss_g = fastdiff_amd.theta_timestep_loss(model, (X[0], audio), dh)
    assert loss_g.grad_fn is not None and abs(loss_g.item() - loss.item()) < 1e-5 * loss.item()
    with pytest.raises(NotImplementedError, match="inference pipeline"):
        model.sample(X[0].clone().requires_grad_(True), [{"t": 0.0, "c_eps": 0.0, "c_div": 1.0, "sigma": 0.0, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": 0}])


def test_phi_loss_on_the_hip_denoiser(gc, sched, monkeypatch):
    """phi_loss (util.py:328-362) with the HIP module as `net` and the stand-in noise_pred of the fixture attached to it: the
    reference's value (golden: gen_phi_loss, steps and z replayed); the stock module (no noise_pred) ends in AttributeError."""
    import fastdiff_amd
    import synth
    from fastdiff_amd import sampler
    g = load_golden("phi_loss")
    monkeypatch.setattr(sampler, "std_normal", lambda size: torch.from_numpy(g["z"].copy()).view(*size).cuda())
    monkeypatch.setattr(torch, "randint", lambda *a, **k: torch.from_numpy(g["ts"].copy()))
    dh = {"T": 1000, "alpha": torch.from_numpy(sched["train_alpha"]).cuda(), "tau": int(g["tau"])}
    X = (torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["audio"]).cuda())
    m = gc.make_model()
    with torch.no_grad():
        with pytest.raises(AttributeError):
            fastdiff_amd.phi_loss(m, X, dh)
        m.noise_pred = synth.stub_noise_pred_batch
        loss = fastdiff_amd.phi_loss(m, X, dh)
    ref = float(g["loss_f64"])
    print("phi loss %.9f, |d| vs f64 reference %.2e (fp32 reference: %.2e)" % (loss.item(), abs(loss.item() - ref), abs(float(g["loss_f32"]) - ref)))
    assert abs(loss.item() - ref) < 2e-6 * abs(ref)


def test_philox_noise_statistics(model):
    import fastdiff_amd
    B, T = 4, 16
    dh = fastdiff_amd.schedules.training_hyperparams()
    rows = [{"t": 0.0, "c_eps": 0.0, "c_div": 1.0, "sigma": 1.0, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": 1},
            {"t": 0.0, "c_eps": 0.0, "c_div": 1.0, "sigma": 0.0, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": 0}]
    with torch.no_grad():
        seq = model.sample(torch.zeros(B, 80, T).cuda(), rows, seed=1234, return_sequence=True)
    x_T, x1 = seq[0].double(), seq[1].double()
    z = x1 - x_T                                  # c_eps = 0: the step only adds sigma*z
    for v in (x_T, z):
        assert abs(v.mean().item()) < 0.02 and abs(v.var().item() - 1.0) < 0.03
        assert abs((v ** 4).mean().item() - 3.0) < 0.2
    assert abs((x_T * z).mean().item()) < 0.02    # independent streams


# ------------------------------------------------------------------------------------------------ full BASELINE size
def test_full_size_forward_against_oracle(model, gc, oracle64):
    """BASELINE shape: one 80x864 mel (10.03 s).  Direct comparison with the fp64 oracle."""
    import synth
    B, T = 1, 864
    mel, audio = synth.synth_mel(1234, B, T), synth.synth_audio(1234, B, T)
    steps = np.array([498.05368], np.float32)
    y_ref = oracle64.forward(audio, mel, steps)
    y = gc.run_forward(model, audio, mel, steps)
    assert gc.maxdiff(y, y_ref) < FWD_TOL


def test_full_size_batch_properties(model, gc):
    """B=8 x T=864 (BASELINE config 2): size-independent properties -- batch items are independent and bit-reproducible,
    the fast set agrees with the naive set, and a one-frame mel perturbation stays inside the finite receptive field
    (+-16 frames)."""
    import synth
    B, T = 8, 864
    mel, audio = synth.synth_mel(99, B, T), synth.synth_audio(99, B, T)
    steps = np.full(B, 74.992283, np.float32)
    y = gc.run_forward(model, audio, mel, steps)
    assert np.isfinite(y).all()
    y_again = gc.run_forward(model, audio, mel, steps)
    assert np.array_equal(y, y_again)
    y3 = gc.run_forward(model, audio[3:4], mel[3:4], steps[3:4])
    assert np.array_equal(y[3:4], y3)
    model.set_option("kernels", "naive")
    try:
        y_naive = gc.run_forward(model, audio[:1], mel[:1], steps[:1])
    finally:
        model.set_option("kernels", "fast")
    assert gc.maxdiff(y[:1], y_naive) < FWD_TOL
    mel2 = mel.copy()
    mel2[0, :, 400] += 1.0
    y2 = gc.run_forward(model, audio, mel2, steps)
    d = np.abs(y2 - y)[0, 0]
    changed = np.nonzero(d > 0)[0]
    assert changed.size > 0
    assert changed.min() >= (400 - 16) * 256 and changed.max() < (400 + 17) * 256
    assert np.array_equal(y2[1:], y[1:])


def test_full_size_sampler_n4(model, gc, sched):
    """N=4, B=8, T=864 through the hipGraph path: finite, reproducible with injected seed, per-item independent."""
    import synth
    B, T = 8, 864
    mel = torch.from_numpy(synth.synth_mel(5, B, T)).cuda()
    rows, _ = gc.table_rows(sched, 4)
    with torch.no_grad():
        a = model.sample(mel, rows, seed=77)
        b = model.sample(mel, rows, seed=77)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert a.shape == (B, 1, T * 256)


@pytest.mark.parametrize("ddim", [False, True])
def test_full_size_sampler_against_oracle(model, gc, sched, oracle64, ddim):
    """BASELINE shape through the whole reverse loop: B=1, T=864 (221,184 samples), N=4, injected x_T and z, against the float64
    oracle's sampling_given_noise_schedule (util.py:158-235), DDPM and "ddim" branches.  Tolerance: the N<=8 loop bar, 1e-4."""
    import synth
    B, T, N = 1, 864, 4
    mel = synth.synth_mel(21, B, T)
    x_T = synth.hash_normal(21, 1, B * T * 256).reshape(B, 1, T * 256)
    z = gc.noise_from_seed(21, B, T, N)
    rows, table = gc.table_rows(sched, N)
    ref = oracle64.sample(mel, table, x_T, z, ddim=ddim)
    with torch.no_grad():
        y = model.sample(torch.from_numpy(mel).cuda(), rows, ddim=ddim, x_T=torch.from_numpy(x_T).cuda(),
                         noise=torch.from_numpy(gc.exec_order_noise(z)).cuda())
    d = gc.maxdiff(y.cpu().numpy(), ref)
    print(f"full-size N=4 sampler (ddim={ddim}): max|d| = {d:.3e}, max|x_0| = {np.abs(ref).max():.3f}")
    assert d < LOOP_TOL
    assert not model.read_tap("range_flags").view(np.int32).any()      # the fp16x2 pipe did this, not the fp32 fallback


def test_full_size_batch_sampler_items_equal_alone_runs(model, gc, sched):
    """B=8, T=864, N=4 on device Philox noise with per-utterance streams: two items of the batch are bit-equal to their own B=1
    runs (so the B=1 oracle comparison above speaks for every item of the benchmark batch)."""
    import synth
    B, T = 8, 864
    mel = torch.from_numpy(synth.synth_mel(5, B, T)).cuda()
    rows, _ = gc.table_rows(sched, 4)
    with torch.no_grad():
        y = model.sample(mel, rows, seed=77, stream_ids=list(range(100, 100 + B)))
        for b in (2, 7):
            alone = model.sample(mel[b:b + 1].contiguous(), rows, seed=77, stream_ids=[100 + b])
            assert torch.equal(y[b], alone[0]), b
        other = model.sample(mel[2:3].contiguous(), rows, seed=77, stream_ids=[107])
        assert not torch.equal(other[0], y[2])                          # the stream id, not the batch position, picks the noise


def device_noise(model, T, N, seed, uid):
    """The exact x_T and z_k the device draws for one utterance of noise stream (seed, uid): the sampler run with c_eps = 0,
    c_div = 1 on a zero mel, so that x never sees the network.  z[k] is the draw of executed step k (zeros where add_noise = 0)."""
    L = T * 256
    idle = {"t": 0.0, "c_eps": 0.0, "c_div": 1.0, "sigma": 0.0, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": 0}
    mel0 = torch.zeros(1, 80, T).cuda()
    with torch.no_grad():
        x_T = model.sample(mel0, [idle] * N, seed=seed, stream_ids=[uid], return_sequence=True)[0].cpu().numpy()
        z = np.zeros((N, 1, 1, L), np.float32)
        for k in range(N - 1):
            rows = [dict(idle) for _ in range(N)]
            rows[k]["sigma"], rows[k]["add_noise"] = 1.0, 1
            z[k] = model.sample(mel0, rows, seed=seed, stream_ids=[uid], x_T=torch.zeros(1, 1, L).cuda()).cpu().numpy()
    return x_T, z


def test_config5_tacotron_batch16_through_the_driver(gc, sched, oracle64):
    """BASELINE configs[4]: 16 Tacotron-range mels (rand*13.5 - 11.5, the range of ln(clamp(., 1e-5))), stored [T, 80] as on disk,
    through infer.synthesize (test-time collater incl. the dropped last frame, one padded batch of 16 with `lens`, device Philox
    noise, device int16 epilogue), N=4.  Three items -- the shortest, the longest, one in between -- against the float64 oracle's
    N=4 output for the SAME noise (read back from the device), peak-normalised to int16: within 1 LSB."""
    from fastdiff_amd import infer
    m = gc.make_model()
    rng = np.random.default_rng(50)
    lens = rng.integers(300, 865, 16).tolist()
    lens[3], lens[11] = 300, 864
    items = [{"item_name": f"taco{i:02d}.npy", "mel": torch.from_numpy((rng.random((t, 80)) * 13.5 - 11.5).astype(np.float32)), "len": t}
             for i, t in enumerate(lens)]
    pcm = infer.synthesize(m, items, n_steps=4, max_batch=16, seed=2024)
    assert sorted(pcm) == sorted(it["item_name"] for it in items)
    rows, table = gc.table_rows(sched, 4)
    worst = 0
    for i in (3, 11, 6):
        t = lens[i] - 1                                         # the collater drops the last frame (dataset_utils.py:116-125)
        assert pcm[items[i]["item_name"]].shape == (t * 256,)
        x_T, z_exec = device_noise(m, t, 4, 2024, i)
        mel = np.ascontiguousarray(items[i]["mel"].numpy()[:t].T)[None]
        ref = oracle64.sample(mel, table, x_T, np.ascontiguousarray(z_exec[::-1]))
        ref16 = oracle64.peak_normalize_int16(ref.astype(np.float32)).reshape(-1)
        d = np.abs(pcm[items[i]["item_name"]].astype(np.int32) - ref16.astype(np.int32))
        worst = max(worst, int(d.max()))
        print(f"config5 item {i} (T={t}): int16 max|d| = {int(d.max())}, differing samples {int((d > 0).sum())} of {d.size}")
        assert d.max() <= 1, i
    assert not m.read_tap("range_flags").view(np.int32).any()


def test_config4_batch64_ragged_n6(gc, sched, oracle64):
    """BASELINE config 4 at one GPU's share and beyond: B=64 zero-padded utterances with T_i ~ U{200..864} (seeded), the N=6
    schedule (FastDiff.py:86-87), `lens` given.  Properties: finite inside every utterance, reproducible, and sampled utterances
    (the shortest, the longest, one in the middle of the batch) bit-identical to running them alone."""
    import synth
    m = gc.make_model()
    B, T = 64, 864
    rng = np.random.default_rng(4)
    lens = rng.integers(200, 865, B).tolist()
    lens[17] = 864
    mel = synth.synth_mel(61, B, T)
    for b, t in enumerate(lens):
        mel[b, :, t:] = 0.0
    rows, _ = gc.table_rows(sched, 6)
    assert len(rows) == 6
    x_T = synth.hash_normal(13, 1, B * T * 256).reshape(B, 1, T * 256)
    melc, xc = torch.from_numpy(mel).cuda(), torch.from_numpy(x_T).cuda()
    with torch.no_grad():
        y = m.sample(melc, rows, x_T=xc, seed=5, lens=lens)
        y2 = m.sample(melc, rows, x_T=xc, seed=5, lens=lens)
        for b, t in enumerate(lens):
            assert torch.isfinite(y[b, :, : t * 256]).all(), b
            assert torch.equal(y[b, :, : t * 256], y2[b, :, : t * 256]), b
    # (without fd_set_noise_streams) Philox noise is indexed by the position in the padded batch, so the alone-runs inject it
    N = len(rows)
    picks = [int(np.argmin(lens)), 17, 40]
    z = np.stack([synth.hash_normal(14, 2 + k, len(picks) * T * 256).reshape(len(picks), 1, T * 256) for k in range(N)])
    zfull = torch.zeros(N, B, 1, T * 256)
    for i, b in enumerate(picks):
        zfull[:, b] = torch.from_numpy(z[:, i])
    with torch.no_grad():
        yb = m.sample(melc, rows, x_T=xc, noise=zfull.cuda(), lens=lens)
        for i, b in enumerate(picks):
            t = lens[b]
            alone = m.sample(melc[b:b + 1, :, :t].contiguous(), rows, x_T=xc[b:b + 1, :, : t * 256].contiguous(),
                             noise=torch.from_numpy(np.ascontiguousarray(z[:, i:i + 1, :, : t * 256])).cuda())
            assert torch.equal(yb[b, :, : t * 256], alone[0]), b
    assert not m.read_tap("range_flags").view(np.int32).any()
    # ... and the shortest and the longest item of the lens-masked batch against the float64 oracle's own N = 6 reverse loop on that
    # utterance alone (util.py:158-235 with the schedule of FastDiff.py:85-87): HIP batch vs oracle, not HIP vs HIP
    _, table = gc.table_rows(sched, 6)
    for i, b in enumerate(picks[:2]):
        t = lens[b]
        ref = oracle64.sample(mel[b:b + 1, :, :t], table, x_T[b:b + 1, :, : t * 256], np.ascontiguousarray(z[::-1][:, i:i + 1, :, : t * 256]))
        d = gc.maxdiff(yb[b:b + 1, :, : t * 256].cpu().numpy(), ref)
        print(f"config4 item {b} (T={t}) of the B=64 ragged N=6 batch vs float64 oracle: max|d| = {d:.3e}, max|x_0| = {np.abs(ref).max():.3f}")
        assert d < LOOP_TOL, (b, d)


def test_long_utterance_beyond_the_benchmark_length(model, gc):
    """One 58 s utterance (T = 5000 frames, 1.28 M samples): the tuned kernels against the one-thread-per-output set, and the
    receptive-field property at the far end (index arithmetic past 2^20 samples per channel, 32 channels -> past 2^25 floats)."""
    import synth
    B, T = 1, 5000
    mel, audio = synth.synth_mel(71, B, T), synth.synth_audio(71, B, T)
    steps = np.array([23.4676], np.float32)
    y = gc.run_forward(model, audio, mel, steps)
    assert np.isfinite(y).all()
    model.set_option("kernels", "naive")
    try:
        y_naive = gc.run_forward(model, audio, mel, steps)
    finally:
        model.set_option("kernels", "fast")
    assert gc.maxdiff(y, y_naive) < FWD_TOL
    mel2 = mel.copy()
    mel2[0, :, 4990] -= 1.0
    d = np.abs(gc.run_forward(model, audio, mel2, steps) - y)[0, 0]
    changed = np.nonzero(d > 0)[0]
    assert changed.size > 0 and changed.min() >= (4990 - 16) * 256


def test_batch_past_two_to_the_31_record_elements(gc):
    """Maximum sizes: 104 utterances x 864 frames put the predicted-kernel records of one block past 2^31 elements (104 * 864 * 24832),
    the activations at 0.74e9 floats per buffer.  Utterances at both ends and in the middle of the batch equal themselves alone, bit for
    bit (the lens contract), so no index on the way wrapped; one utterance more than the documented limit (B * T * 256 * 32 < 2^31) is
    refused with the library's message instead of being computed wrongly."""
    import synth
    from fastdiff_amd import infer
    m = gc.make_model()
    rows = infer._step_rows(m, 4, None, None)
    B, T = 104, 864
    mel = torch.from_numpy(synth.synth_mel(5, 4, T)).cuda().repeat(B // 4, 1, 1).contiguous()
    lens = [T - (b % 7) * 3 for b in range(B)]
    with torch.no_grad():
        y = m.sample(mel, rows, seed=9, lens=lens, stream_ids=list(range(B)))
        for b in (0, 51, B - 1):
            y1 = m.sample(mel[b:b + 1, :, : lens[b]].contiguous(), rows, seed=9, stream_ids=[b])
            assert torch.equal(y[b, :, : lens[b] * 256], y1[0]), b
        assert torch.isfinite(y1).all()
        del y, mel
        big = torch.zeros(304, 80, T, device="cuda")
        with pytest.raises(AssertionError, match="too large"):
            m.sample(big, rows, seed=9)
    del m, big
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ epilogue (8f row 1)
def test_peak_normalize_int16_bit_exact(model, oracle64):
    rng = np.random.default_rng(3)
    wav = (rng.standard_normal((3, 1, 4096)) * rng.uniform(0.1, 5.0, (3, 1, 1))).astype(np.float32)
    pcm = model.peak_normalize_int16(torch.from_numpy(wav).cuda()).cpu().numpy()
    ref = oracle64.peak_normalize_int16(wav).reshape(3, 4096)
    assert np.array_equal(pcm, ref)


def test_peak_normalize_int16_ragged_equals_each_utterance_alone(model, oracle64):
    """fd_peak_normalize_int16_ragged: the peak of every utterance of a zero-padded batch is searched over its own samples only
    (garbage, even NaN, behind them must not matter) and the PCM behind them is silence."""
    rng = np.random.default_rng(4)
    L, valid = 5000, [5000, 1, 2048, 3333]
    wav = (rng.standard_normal((4, 1, L)) * rng.uniform(0.1, 5.0, (4, 1, 1))).astype(np.float32)
    dirty = wav.copy()
    for b, n in enumerate(valid):
        dirty[b, 0, n:] = np.nan if b % 2 else 1.0e9
    pcm = model.peak_normalize_int16(torch.from_numpy(dirty).cuda(), valid=valid).cpu().numpy()
    for b, n in enumerate(valid):
        ref = oracle64.peak_normalize_int16(wav[b:b + 1, :, :n]).reshape(n)
        assert np.array_equal(pcm[b, :n], ref), b
        assert not pcm[b, n:].any(), b
    with pytest.raises(Exception, match="valid"):
        model.peak_normalize_int16(torch.from_numpy(wav).cuda(), valid=[5000, 0, 1, 1])


def test_ragged_batch_is_refused_on_the_naive_kernels(gc):
    """The straightforward kernels compute the padded tensor: with `lens` they would silently ignore the lengths (results then depend on
    what the padding region of the workspace holds).  The library refuses instead; full-length `lens` are no ragged batch."""
    import synth
    m = gc.make_model()
    B, T = 2, 20
    mel = torch.from_numpy(synth.synth_mel(3, B, T)).cuda()
    rows = [{"t": 5.0, "c_eps": 0.1, "c_div": 1.0, "sigma": 0.0, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": 0}]
    m.set_option("kernels.lvc", "naive")
    with torch.no_grad():
        m.sample(mel, rows, seed=1, lens=[T, T])
        with pytest.raises(NotImplementedError, match="ragged batch"):
            m.sample(mel, rows, seed=1, lens=[T, T - 3])
    m.set_option("kernels.lvc", "fast")
    with torch.no_grad():
        assert torch.isfinite(m.sample(mel, rows, seed=1, lens=[T, T - 3])[1, 0, : (T - 3) * 256]).all()
