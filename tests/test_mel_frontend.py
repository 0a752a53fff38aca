"""Mel front-end: the restated librosa calls of process_utterance (data_gen_utils.py:93-147).

librosa is not installable here, so the pin is split: the STFT restatement is checked against torch.stft (an independent
implementation of the same definition); the Slaney filterbank is checked for the structure librosa documents (the oracle header
says "parity unpinned" for it); the HIP kernel is checked against the float64 oracle on noise, a tone and a real recording."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mel_frontend as mf   # noqa: E402


def test_stft_restatement_matches_torch_stft():
    rng = np.random.default_rng(0)
    for n in (256, 1000, 5000, 41728):
        wav = rng.standard_normal(n) * 0.1
        ours = mf.stft_mag(wav)
        ref = torch.stft(torch.from_numpy(wav), 1024, 256, 1024, torch.hann_window(1024, periodic=True, dtype=torch.float64),
                         center=True, pad_mode="constant", return_complex=True).abs().numpy()
        assert ours.shape == ref.shape == (513, 1 + n // 256)
        assert np.abs(ours - ref).max() < 1e-12


def test_mel_filterbank_structure():
    B = mf.mel_basis()
    assert B.shape == (80, 513) and (B >= 0).all()
    freqs = np.linspace(0, 11025, 513)
    edges = mf.mel_to_hz(np.linspace(mf.hz_to_mel(80.0), mf.hz_to_mel(7600.0), 82))
    assert abs(edges[0] - 80.0) < 1e-9 and abs(edges[-1] - 7600.0) < 1e-6 and (np.diff(edges) > 0).all()
    assert abs(mf.hz_to_mel(1000.0) - 15.0) < 1e-12 and abs(mf.mel_to_hz(mf.hz_to_mel(4321.0)) - 4321.0) < 1e-9      # Slaney scale
    for m in range(80):
        nz = np.nonzero(B[m])[0]
        assert len(nz) >= 1 and (np.diff(nz) == 1).all()                                   # one contiguous triangle
        assert edges[m] < freqs[nz[0]] and freqs[nz[-1]] < edges[m + 2]                     # inside its own band
        peak = freqs[nz[np.argmax(B[m, nz])]]
        assert abs(peak - edges[m + 1]) <= 11025 / 512                                      # apex at the centre frequency (bin grid)
        assert B[m].max() <= 2.0 / (edges[m + 2] - edges[m]) + 1e-12                        # norm="slaney": height 2 / bandwidth
    wide = [m for m in range(80) if np.count_nonzero(B[m]) >= 12]
    area = B[wide].sum(1) * (11025 / 512)                                                   # ... so every filter has unit area
    assert np.abs(area - 1.0).max() < 0.02
    assert B[:, freqs < 80].sum() == 0 and B[:, freqs > 7600].sum() == 0


def test_tone_lands_in_the_right_filter():
    t = np.arange(22050) / 22050.0
    for f0 in (220.0, 1000.0, 3500.0):
        m = mf.log_mel(0.5 * np.sin(2 * np.pi * f0 * t))[:, 40]
        edges = mf.mel_to_hz(np.linspace(mf.hz_to_mel(80.0), mf.hz_to_mel(7600.0), 82))
        k = int(np.argmax(m))
        assert edges[k] <= f0 <= edges[k + 2]


def test_golden_fixture_is_what_the_oracle_gives():
    g = load_golden("frontend_lj001_0002")
    mel = mf.log_mel(g["pcm"].astype(np.float64) / 32768.0)
    assert mel.shape == (80, 164) and np.abs(mel - g["mel_f64"]).max() < 1e-12


def test_tacotron_restatement_matches_the_reference_classes():
    """frontend_tacotron_lj001_0002 was produced by EXECUTING the reference's TacotronSTFT / STFT classes (oracle/gen_golden.py
    gen_frontend_tacotron; float32 conv1d DFT) on the sample recording: the float64 restatement must agree to float32 rounding."""
    g, t = load_golden("frontend_lj001_0002"), load_golden("frontend_tacotron_lj001_0002")
    ours = mf.tacotron_log_mel(g["pcm"].astype(np.float64) / 32768.0)
    ref = t["mel_ref_f32"].astype(np.float64)
    assert ours.shape == ref.shape == (80, 164)
    assert np.abs(np.exp(ours) - np.exp(ref)).max() < 1e-5 * np.exp(ref).max()           # linear domain, relative to the loudest bin
    loud = ref > -7.0                                                                      # mel > 1e-3, two decades above the 1e-5 clamp
    assert np.abs(ours - ref)[loud].max() < 2e-4
    # reflect padding really differs from zero padding at the edges, and only there
    a, b = mf.stft_mag_reflect(g["pcm"] / 32768.0), mf.stft_mag(g["pcm"] / 32768.0)
    assert np.abs(a - b)[:, 2:-3].max() < 1e-12 and np.abs(a - b)[:, :2].max() > 1e-6


def test_pwg_restatement_matches_the_references_process_utterance():
    """frontend_pwg_lj001_0002 was produced by EXECUTING the reference's process_utterance (data_gen_utils.py:93-147, cut out with
    ast; oracle/gen_golden.py gen_frontend_pwg) on the sample recording, its librosa.stft served by torch.stft and its
    librosa.filters.mel by the restated bank -- so this pins everything of the 'pwg' front-end except the filter values themselves
    (magnitude, matrix product, log10 clamp at 1e-6, frame count, returned-wav length): the restatement must be that function."""
    g, r = load_golden("frontend_lj001_0002"), load_golden("frontend_pwg_lj001_0002")
    ours = mf.log_mel(g["pcm"].astype(np.float64) / 32768.0)
    assert ours.shape == r["mel_ref_f64"].shape == (80, 164)
    assert np.abs(ours - r["mel_ref_f64"]).max() < 1e-6                  # float64 run of the reference function (mel_basis is float32 there)
    loud = r["mel_ref_f64"] > -4.0
    assert np.abs(ours - r["mel_ref_f32"])[loud].max() < 2e-4            # float32 run: what librosa.core.load + librosa.stft would feed it
    assert int(r["wav_len_f64"]) == 164 * 256                            # wav padded to the frame grid and trimmed (data_gen_utils.py:137-139)


@pytest.mark.gpu
def test_device_pwg_front_end_matches_the_references_process_utterance():
    import fastdiff_amd
    torch.manual_seed(1234)
    model = fastdiff_amd.FastDiff().cuda().eval()
    g, r = load_golden("frontend_lj001_0002"), load_golden("frontend_pwg_lj001_0002")
    got = model.mel_spectrogram(torch.from_numpy(g["pcm"].astype(np.float32) / 32768.0).cuda())[0].cpu().numpy().astype(np.float64)
    ref = r["mel_ref_f64"]
    assert got.shape == ref.shape
    assert np.abs(got - ref)[ref > -4.0].max() < 2e-4
    assert np.abs(10.0 ** got - 10.0 ** ref).max() < 2e-6 * max(1.0, float((10.0 ** ref).max()))


@pytest.mark.gpu
def test_device_tacotron_front_end_matches_the_oracle():
    import fastdiff_amd
    torch.manual_seed(1234)
    model = fastdiff_amd.FastDiff().cuda().eval()
    g, t = load_golden("frontend_lj001_0002"), load_golden("frontend_tacotron_lj001_0002")
    rng = np.random.default_rng(6)
    cases = {"speech": g["pcm"].astype(np.float32) / 32768.0, "noise": (rng.standard_normal(7777) * 0.05).astype(np.float32),
             "short": (rng.standard_normal(513) * 0.1).astype(np.float32)}
    for name, wav in cases.items():
        ref = mf.tacotron_log_mel(wav.astype(np.float64))
        got = model.mel_spectrogram(torch.from_numpy(wav).cuda(), variant="tacotron")[0].cpu().numpy()
        assert got.shape == ref.shape
        lin = np.abs(np.exp(got.astype(np.float64)) - np.exp(ref))
        loud = ref > -7.0
        print(name, "max |d ln mel| loud bins %.2e, max |d mel| %.2e" % (np.abs(got - ref)[loud].max(), lin.max()))
        assert np.abs(got - ref)[loud].max() < 5e-4 and lin.max() < 2e-6 * max(1.0, float(np.exp(ref).max()))
    # against what the reference's own classes produced for the recording (float32 direct DFT there)
    got = model.mel_spectrogram(torch.from_numpy(cases["speech"]).cuda(), variant="tacotron")[0].cpu().numpy()
    assert np.abs(np.exp(got.astype(np.float64)) - np.exp(t["mel_ref_f32"].astype(np.float64))).max() < 1e-5 * float(np.exp(t["mel_ref_f32"]).max())
    # the default variant is untouched by the option, and the reference's checks are mirrored
    pwg = model.mel_spectrogram(torch.from_numpy(cases["speech"]).cuda())[0].cpu().numpy()
    assert np.abs(pwg - mf.log_mel(cases["speech"].astype(np.float64)))[mf.log_mel(cases["speech"].astype(np.float64)) > -4.0].max() < 2e-4
    with pytest.raises(AssertionError):
        model.mel_spectrogram(torch.full((1, 4000), 1.5).cuda(), variant="tacotron")       # layers.py:70-71
    with pytest.raises(Exception, match="reflect"):
        model.mel_spectrogram(torch.zeros(1, 512).cuda(), variant="tacotron")              # F.pad(reflect) needs pad < length
    with pytest.raises(ValueError):
        model.mel_spectrogram(torch.zeros(1, 4000).cuda(), variant="hifigan")


@pytest.mark.gpu
def test_device_front_end_matches_the_oracle():
    import fastdiff_amd
    torch.manual_seed(1234)
    model = fastdiff_amd.FastDiff().cuda().eval()
    g = load_golden("frontend_lj001_0002")
    rng = np.random.default_rng(5)
    t = np.arange(30000) / 22050.0
    cases = {"speech": g["pcm"].astype(np.float32) / 32768.0,
             "noise": (rng.standard_normal(12345) * 0.05).astype(np.float32),
             "tone+silence": np.concatenate([0.3 * np.sin(2 * np.pi * 440.0 * t), np.zeros(5000)]).astype(np.float32)}
    for name, wav in cases.items():
        ref = mf.log_mel(wav.astype(np.float64))
        f32 = mf.log_mel(wav, np.float32)                                                  # the same algorithm in float32 (numpy FFT)
        got = model.mel_spectrogram(torch.from_numpy(wav).cuda())[0].cpu().numpy()
        assert got.shape == ref.shape
        err, err32 = np.abs(got - ref), np.abs(f32 - ref)
        loud = ref > -4.0                                                                  # mel > 1e-4: well above the 1e-6 floor
        lin = np.abs(10.0 ** got.astype(np.float64) - 10.0 ** ref)                          # quiet bins: judged on the mel itself (floor 1e-6)
        print(name, "max |d log10 mel|: device %.2e (loud bins %.2e), float32 numpy %.2e; max |d mel| %.2e" % (err.max(), err[loud].max(), err32.max(), lin.max()))
        assert err[loud].max() < 2e-4 and lin.max() < 2e-6 * max(1.0, float((10.0 ** ref).max()))
    # batch of two with a shorter frame count, and the argument checks
    two = torch.from_numpy(np.stack([cases["noise"][:8000], cases["tone+silence"][:8000]])).cuda()
    m2 = model.mel_spectrogram(two, n_frames=20).cpu().numpy()
    assert np.abs(m2[0] - mf.log_mel(cases["noise"][:8000].astype(np.float64))[:, :20]).max() < 2e-4
    with pytest.raises(Exception, match="fd_mel_spectrogram"):
        model.mel_spectrogram(two, n_frames=40)
