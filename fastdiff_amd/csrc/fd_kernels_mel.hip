// fd_kernels_mel.hip -- the mel front-end in front of the vocoder: wav -> log-mel [80, T].
//
// Reference: data_gen/tts/data_gen_utils.py:93-147 (process_utterance, vocoder='pwg'):
//   librosa.stft(n_fft=1024, hop=256, win=1024, window="hann", center, pad_mode="constant") -> |.| -> librosa.filters.mel(22050, 1024,
//   80, 80, 7600) @ . -> log10(max(1e-6, .)).   One workgroup = one frame.  The 1024-point DFT is done as 32 x 32 (Cooley-Tukey, one
// split): n = 32 n1 + n2, k = k1 + 32 k2,
//     A[n2][k1] = sum_n1 x[32 n1 + n2] W32^(n1 k1);   B = A * W1024^(n2 k1);   X[k1 + 32 k2] = sum_n2 B[n2][k1] W32^(n2 k2)
// two passes of 32-term sums through LDS (every table index is an exact integer product mod 32 / 1024; no recurrences), ~700 FMAs per
// thread instead of the ~4000 LDS-bound ones of a direct DFT (0.16 ms for the B=8 x 864-frame benchmark batch).  The 513 magnitudes go
// back to LDS and 80 threads apply their triangular filter and the log.
//
// TACO = the Tacotron front-end (data_gen/tts/tacotron/layers.py:42-80 TacotronSTFT.mel_spectrogram over tacotron/stft.py:78-104
// STFT.transform, driven by vocoder_binarizer_tacotron.py:110-116): the same transform with the signal REFLECT-padded by 512
// (stft.py:84-88), filters.mel(22050, 1024, 80, 0, 8000) and ln(clamp(., 1e-5)) (audio_processing.py:78-84).
#include "fd_internal.h"
#include "fd_kernels.h"

namespace fdk {

template <bool TACO>
__global__ void __launch_bounds__(256) k_mel_frontend(const float *__restrict__ wav, float *__restrict__ mel, const float *__restrict__ tab,
                                                      const int *__restrict__ fb_lo, const int *__restrict__ fb_n,
                                                      const int *__restrict__ fb_off, const float *__restrict__ fb_w, int64_t n_samples,
                                                      int T)
{
    __shared__ float ct[1024], st[1024], xw[1024], br[1024], bi[1024], mag[544];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lo5 = tid & 31, g = tid >> 5;
    const float *w = wav + (int64_t)b * n_samples;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = j * 256 + tid;
        int64_t p = (int64_t)t * 256 + n - 512;                       // center=True: n_fft/2 zeros in front (pad_mode="constant")
        if (TACO) p = p < 0 ? -p : (p >= n_samples ? 2 * (n_samples - 1) - p : p);      // ... or the signal mirrored about its end samples
        ct[n] = tab[n];
        st[n] = tab[1024 + n];
        xw[n] = (p >= 0 && p < n_samples) ? w[p] * tab[2048 + n] : 0.0f;      // periodic Hann window
    }
    __syncthreads();
    // pass 1: thread = (n2 = lo5, k1 = 4g .. 4g+3); W32^m = (ct, -st)[32 m]
    {
        float re[4] = {0.f, 0.f, 0.f, 0.f}, im[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int n1 = 0; n1 < 32; ++n1) {
            const float x = xw[32 * n1 + lo5];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = ((n1 * (4 * g + j)) & 31) << 5;
                re[j] = fmaf(x, ct[m], re[j]);
                im[j] = fmaf(-x, st[m], im[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                  // twiddle W1024^(n2 k1) = (c, -s)
            const int k1 = 4 * g + j, m = lo5 * k1;                     // <= 31 * 31
            const float c = ct[m], sn = st[m];
            br[lo5 * 32 + k1] = re[j] * c + im[j] * sn;
            bi[lo5 * 32 + k1] = im[j] * c - re[j] * sn;
        }
    }
    __syncthreads();
    // pass 2: thread = (k1 = lo5, k2 in {g, g + 8, 16 (g == 0)}); only k <= 512 is kept
    {
        float re[3] = {0.f, 0.f, 0.f}, im[3] = {0.f, 0.f, 0.f};
        const int k2s[3] = {g, g + 8, 16};
#pragma unroll 8
        for (int n2 = 0; n2 < 32; ++n2) {
            const float xr = br[n2 * 32 + lo5], xi = bi[n2 * 32 + lo5];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int m = ((n2 * k2s[j]) & 31) << 5;
                const float c = ct[m], sn = st[m];                       // times (c - i s)
                re[j] = fmaf(xr, c, fmaf(xi, sn, re[j]));
                im[j] = fmaf(xi, c, fmaf(-xr, sn, im[j]));
            }
        }
        mag[lo5 + 32 * g] = sqrtf(re[0] * re[0] + im[0] * im[0]);
        mag[lo5 + 32 * (g + 8)] = sqrtf(re[1] * re[1] + im[1] * im[1]);
        if (g == 0 && lo5 == 0) mag[512] = sqrtf(re[2] * re[2] + im[2] * im[2]);
    }
    __syncthreads();
    if (tid < 80) {
        const float *wv = fb_w + fb_off[tid];
        const float *mg = mag + fb_lo[tid];
        float acc = 0.0f;
        for (int j = 0; j < fb_n[tid]; ++j) acc = fmaf(wv[j], mg[j], acc);
        mel[((int64_t)b * 80 + tid) * T + t] = TACO ? logf(fmaxf(1e-5f, acc)) : log10f(fmaxf(1e-6f, acc));
    }
}

hipError_t mel_frontend(const Launch &L, const float *wav, int B, int64_t n_samples, float *mel, int T)
{
    const MelTables &m = L.ctx->mel[L.ctx->mel_variant];
    if (L.ctx->mel_variant == MEL_TACOTRON)
        FD_LAUNCH(L, "mel_frontend_tacotron", k_mel_frontend<true>, dim3(T, B), dim3(256), 0, wav, mel, m.tab, m.fb_lo, m.fb_n, m.fb_off, m.fb_w, n_samples, T);
    else
        FD_LAUNCH(L, "mel_frontend", k_mel_frontend<false>, dim3(T, B), dim3(256), 0, wav, mel, m.tab, m.fb_lo, m.fb_n, m.fb_off, m.fb_w, n_samples, T);
    return hipSuccess;
}

}  // namespace fdk
