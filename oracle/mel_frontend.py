"""CPU restatement of the reference's mel front-end -- TEST INFRASTRUCTURE.

Reference: data_gen/tts/data_gen_utils.py:93-147 (`process_utterance`, vocoder='pwg'):
    x_stft = librosa.stft(wav, n_fft=1024, hop_length=256, win_length=1024, window="hann", pad_mode="constant")
    mel    = librosa.filters.mel(22050, 1024, 80, 80, 7600) @ np.abs(x_stft)
    mel    = np.log10(np.maximum(1e-6, mel))                                      # [80, T], T = 1 + len(wav) // 256
librosa is not installed in this image and cannot be fetched, so the two librosa calls are restated from their published
definitions (librosa 0.8/0.9, the era of the reference's requirements.txt):
  * stft: centered frames (n_fft // 2 zeros on both sides with pad_mode="constant"), periodic Hann window
    (scipy.signal.get_window("hann", 1024, fftbins=True)), X[k, t] = sum_n w[n] y[t*hop + n] exp(-2 pi i k n / n_fft), k <= n_fft/2.
    PINNED here against torch.stft, an independent implementation of the same definition (tests/test_mel_frontend.py).
  * filters.mel: Slaney mel scale (htk=False: linear below 1 kHz with 200/3 Hz per mel, logarithmic above with step ln(6.4)/27),
    n_mels + 2 equally spaced mel points between fmin and fmax, triangular weights on the FFT bin centres, each filter scaled by
    2 / (f[m+2] - f[m]) (norm="slaney").  PARITY UNPINNED against librosa itself: only structural properties are tested.
"""
import numpy as np

SR, N_FFT, HOP, N_MELS, FMIN, FMAX, EPS = 22050, 1024, 256, 80, 80.0, 7600.0, 1e-6


def hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, min_log_hz) / min_log_hz) / logstep, mel)


def mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX):
    """[n_mels, n_fft//2 + 1] float64 (librosa returns float32 of the same numbers)."""
    fftfreqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    return weights * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


def hann_periodic(n=N_FFT):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_mag(wav, dtype=np.float64):
    """|STFT| [n_fft//2 + 1, T], T = 1 + len(wav) // hop."""
    y = np.pad(np.asarray(wav, dtype), N_FFT // 2)
    T = 1 + (len(y) - N_FFT) // HOP
    frames = np.lib.stride_tricks.as_strided(y, (T, N_FFT), (y.strides[0] * HOP, y.strides[0]))
    return np.abs(np.fft.rfft(frames * hann_periodic().astype(dtype), axis=1)).T


def log_mel(wav, dtype=np.float64):
    """[80, T] log10(max(1e-6, mel_basis @ |STFT|))."""
    return np.log10(np.maximum(EPS, mel_basis().astype(dtype) @ stft_mag(wav, dtype)))


# ---- the Tacotron front-end -------------------------------------------------------------------------------------------------------
# Reference: data_gen/tts/tacotron/layers.py:42-80 (TacotronSTFT, built at vocoder_binarizer_tacotron.py:44-47 with fft 1024, hop 256,
# win 1024, 80 bins, 22050 Hz, mel_fmin 0, mel_fmax 8000 -- FastDiff_tacotron.yaml:20-21) over tacotron/stft.py:41-104 (STFT):
#     y reflect-padded by n_fft // 2 (stft.py:84-88); conv1d with the windowed DFT basis, stride hop (:90-94) = the same centered STFT;
#     magnitude = sqrt(re^2 + im^2) (:100); mel = mel_basis @ magnitude (layers.py:77); log(clamp(mel, min=1e-5)) (audio_processing.py:78-84).
# PINNED by executing those reference classes (oracle/gen_golden.py gen_frontend_tacotron, with stand-ins for the two absent librosa
# helpers: pad_center is the identity for win == n_fft, filters.mel is mel_basis() above -- so the filter bank itself stays unpinned).
TACO_FMIN, TACO_FMAX, TACO_CLIP = 0.0, 8000.0, 1e-5


def stft_mag_reflect(wav, dtype=np.float64):
    y = np.pad(np.asarray(wav, dtype), N_FFT // 2, mode="reflect")
    T = 1 + (len(y) - N_FFT) // HOP
    frames = np.lib.stride_tricks.as_strided(y, (T, N_FFT), (y.strides[0] * HOP, y.strides[0]))
    return np.abs(np.fft.rfft(frames * hann_periodic().astype(dtype), axis=1)).T


def tacotron_log_mel(wav, dtype=np.float64):
    """[80, T] ln(max(1e-5, mel_basis(fmin 0, fmax 8000) @ |STFT of the reflect-padded signal|))."""
    return np.log(np.maximum(TACO_CLIP, mel_basis(fmin=TACO_FMIN, fmax=TACO_FMAX).astype(dtype) @ stft_mag_reflect(wav, dtype)))
