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


def _independent_mel_bank(sr, n_fft, n_mels, fmin, fmax):
    """A second derivation of librosa.filters.mel(norm="slaney", htk=False) that shares no expression with oracle/mel_frontend.py or
    the library's default_mel_bank: the band edges are found by INVERTING the forward scale numerically (bisection on hz -> mel, no
    closed-form mel -> hz), each filter is the piecewise-linear hat through (f[m], 0), (f[m+1], 1), (f[m+2], 0) evaluated with
    np.interp, and its height is fixed by the analytic unit-area condition of that hat (area = base / 2 -> height 2 / base)."""
    def hz2mel(f):                                     # Slaney's scale as Auditory Toolbox defines it: 3 mel per 200 Hz up to 1 kHz,
        return 3.0 * f / 200.0 if f < 1000.0 else 15.0 + 27.0 * np.log(f / 1000.0) / np.log(6.4)      # then 27 mel per factor 6.4

    def mel2hz(m):
        lo, hi = 0.0, 1.0e5
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            lo, hi = (mid, hi) if hz2mel(mid) < m else (lo, mid)
        return 0.5 * (lo + hi)
    m0, m1 = hz2mel(fmin), hz2mel(fmax)
    edges = np.array([mel2hz(m0 + (m1 - m0) * i / (n_mels + 1)) for i in range(n_mels + 2)])
    bins = np.arange(n_fft // 2 + 1) * (sr / n_fft)
    fb = np.zeros((n_mels, len(bins)))
    for m in range(n_mels):
        hat = np.interp(bins, edges[m:m + 3], [0.0, 1.0, 0.0], left=0.0, right=0.0)
        fb[m] = hat * (2.0 / (edges[m + 2] - edges[m]))
    return fb


def test_mel_filterbank_against_an_independent_derivation():
    for args in ((22050, 1024, 80, 80.0, 7600.0), (22050, 1024, 80, 0.0, 8000.0), (22050, 2048, 128, 0.0, 11025.0)):
        a, b = mf.mel_basis(*args), _independent_mel_bank(*args)
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 1e-12 * b.max() + 1e-15, args
        assert (np.nonzero(a)[0] == np.nonzero(b)[0]).all() and (np.nonzero(a)[1] == np.nonzero(b)[1]).all()


def test_mel_filterbank_against_a_third_party_implementation_of_librosas():
    """Round 6: librosa itself is not in the image, but Hugging Face transformers is, and its `audio_utils.mel_filter_bank(norm="slaney",
    mel_scale="slaney")` is an independent implementation of `librosa.filters.mel` (documented and tested upstream as matching it).
    `tests/golden/mel_bank_third_party.npz` holds what it returns for the two front-ends' arguments (oracle/gen_golden.py mel_bank): the
    restated default banks must equal it to float64 rounding, zero pattern included -- and, where transformers is importable, the
    fixture must be what that library returns here."""
    g = load_golden("mel_bank_third_party")
    for name, band in (("pwg", (80.0, 7600.0)), ("tacotron", (0.0, 8000.0))):
        ref, ours = g[name], mf.mel_basis(fmin=band[0], fmax=band[1])
        assert ref.shape == (80, 513) and ref.dtype == np.float64
        assert np.array_equal(ours != 0, ref != 0)
        assert np.abs(ours - ref).max() <= 1e-14 * ref.max(), (name, float(np.abs(ours - ref).max()))
    try:
        from transformers.audio_utils import mel_filter_bank
    except Exception:      # noqa: BLE001 -- the fixture is the pin; regenerating it is a bonus
        return
    for name, band in (("pwg", (80.0, 7600.0)), ("tacotron", (0.0, 8000.0))):
        fb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=band[0], max_frequency=band[1], sampling_rate=22050,
                             norm="slaney", mel_scale="slaney")
        assert np.abs(np.asarray(fb, np.float64).T - g[name]).max() <= 1e-15


def test_mel_scale_and_filterbank_values_printed_in_librosas_documentation():
    """The numbers librosa's own docstrings print (librosa 0.8 - 0.10: `mel_frequencies`, `hz_to_mel`, `mel_to_hz`, `filters.mel`),
    quoted here as the only librosa-originated values available offline.  They pin the Slaney scale completely (40 band edges to
    1e-3 Hz) and the filter normalisation to the digits the docstring shows."""
    doc_mel_frequencies_40 = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49,
                              1024.856, 1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47,
                              2697.686, 2945.799, 3216.731, 3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92,
                              7101.009, 7754.107, 8467.272, 9246.028, 10096.408, 11025.]      # librosa.mel_frequencies(n_mels=40)
    ours = mf.mel_to_hz(np.linspace(mf.hz_to_mel(0.0), mf.hz_to_mel(11025.0), 40))
    assert np.abs(ours - np.array(doc_mel_frequencies_40)).max() < 5.1e-4
    assert abs(float(mf.hz_to_mel(60.0)) - 0.9) < 1e-12                                    # librosa.hz_to_mel(60) -> 0.9
    assert np.abs(mf.hz_to_mel([110.0, 220.0, 440.0]) - [1.65, 3.3, 6.6]).max() < 1e-12    # -> array([1.65, 3.3, 6.6])
    assert abs(float(mf.mel_to_hz(3.0)) - 200.0) < 1e-12                                   # librosa.mel_to_hz(3) -> 200.
    assert np.abs(mf.mel_to_hz([1, 2, 3, 4, 5]) - [66.667, 133.333, 200.0, 266.667, 333.333]).max() < 5.1e-4
    # librosa.filters.mel(sr=22050, n_fft=2048) prints [[0., 0.016, ..., 0., 0.], ...]; with fmax=8000: [[0., 0.02, ..., 0., 0.], ...]
    full, clipped = mf.mel_basis(22050, 2048, 128, 0.0, 11025.0), mf.mel_basis(22050, 2048, 128, 0.0, 8000.0)
    assert full.shape == (128, 1025) and full[0, 0] == 0.0 and round(float(full[0, 1]), 3) == 0.016 and full[0, -1] == 0.0 and (full[-1, :2] == 0.0).all()
    assert round(float(clipped[0, 1]), 2) == 0.02 and clipped[0, 0] == 0.0 and (clipped[:, -1] == 0.0).all()


@pytest.mark.gpu
def test_library_default_filter_banks_are_the_restated_ones_and_a_supplied_bank_is_used_bit_for_bit():
    """fd_get / fd_set_mel_filterbank (include/fastdiff_hip.h; reference: data_gen_utils.py:122-134, tacotron/layers.py:42-60).  The default
    banks are the oracle's restatement in float32; a bank handed in by the caller is what the kernel applies, value for value: with
    two filter rows exchanged the output rows are exchanged and nothing else changes (torch.equal), scaled by a power of two the
    linear mel scales by exactly that, and NULL restores the default."""
    import fastdiff_amd
    torch.manual_seed(1234)
    model = fastdiff_amd.FastDiff().cuda().eval()
    g = load_golden("frontend_lj001_0002")
    wav = torch.from_numpy(g["pcm"].astype(np.float32) / 32768.0).cuda()
    for variant, band in (("pwg", (80.0, 7600.0)), ("tacotron", (0.0, 8000.0))):
        bank, user = model.mel_filterbank(variant)
        ref = mf.mel_basis(fmin=band[0], fmax=band[1])
        assert not user and bank.shape == (80, 513) and bank.dtype == np.float32
        assert np.array_equal(bank != 0, ref != 0)
        assert np.abs(bank.astype(np.float64) - ref).max() <= 6e-8 * ref.max()            # float32 rounding of the same numbers
        assert np.abs(bank.astype(np.float64) - _independent_mel_bank(22050, 1024, 80, *band)).max() <= 6e-8 * ref.max()
        third = load_golden("mel_bank_third_party")[variant]                                  # transformers' implementation of librosa.filters.mel
        assert np.array_equal(bank != 0, third != 0) and np.abs(bank.astype(np.float64) - third).max() <= 6e-8 * third.max()
        base = model.mel_spectrogram(wav, variant=variant)
        # (1) the same matrix handed back: identical output, and the library says whose it is
        model.set_mel_filterbank(bank, variant)
        got, user = model.mel_filterbank(variant)
        assert user and np.array_equal(got, bank)
        assert torch.equal(model.mel_spectrogram(wav, variant=variant), base)
        # (2) rows 7 and 55 exchanged: the output rows are exchanged, every other value keeps its bits
        perm = bank.copy()
        perm[[7, 55]] = perm[[55, 7]]
        model.set_mel_filterbank(perm, variant)
        out = model.mel_spectrogram(wav, variant=variant)
        idx = list(range(80)); idx[7], idx[55] = 55, 7
        assert torch.equal(out, base[:, idx])
        assert np.array_equal(model.mel_filterbank(variant)[0], perm)
        # (3) every weight doubled: the mel doubles exactly, i.e. its log moves by log(2) in the front-end's own base
        model.set_mel_filterbank(2.0 * bank, variant)
        out2 = model.mel_spectrogram(wav, variant=variant).cpu().numpy().astype(np.float64)
        b = base.cpu().numpy().astype(np.float64)
        floor = -6.0 if variant == "pwg" else np.log(1e-5)
        loud = b > floor + 1.0
        step = np.log10(2.0) if variant == "pwg" else np.log(2.0)
        assert np.abs(out2 - b - step)[loud].max() < 2e-6 * max(1.0, np.abs(b).max())
        # (4) back to the default
        model.set_mel_filterbank(None, variant)
        assert not model.mel_filterbank(variant)[1]
        assert torch.equal(model.mel_spectrogram(wav, variant=variant), base)
    # the other front-end's bank was never touched by the calls on the first, and a malformed bank is refused
    with pytest.raises(ValueError):
        model.set_mel_filterbank(np.zeros((80, 512), np.float32))
    bad = model.mel_filterbank("pwg")[0].copy()
    bad[3, 40] = np.nan
    with pytest.raises(AssertionError, match="not finite"):
        model.set_mel_filterbank(bad)
    # a supplied bank follows the module to a new handle
    model.set_mel_filterbank(perm, "tacotron")
    model._release()
    assert model.mel_filterbank("tacotron")[1] and np.array_equal(model.mel_filterbank("tacotron")[0], perm)
