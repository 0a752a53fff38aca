/*
 * fastdiff_hip_ext.h -- libfastdiff_hip.so: the rows next to the inference path (the int16 waveform epilogue
 * behind fd_sample and the mel front-end in front of it) and the test / introspection hooks.  Conventions: fastdiff_hip.h.
 */
#ifndef FASTDIFF_HIP_EXT_H
#define FASTDIFF_HIP_EXT_H

#include "fastdiff_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Waveform epilogue: wav/abs(wav).max() per utterance (FastDiff.py:110), *32767 -> int16
 * (utils/audio.py:11-16).  wav [B,1,L] device -> pcm [B,L] device int16. */
FD_API int fd_peak_normalize_int16(fd_handle h, const float *wav, int B, int64_t L, int16_t *pcm, void *stream);
/* The same for a zero-padded batch (fd_sample with lens): valid [B] host = samples of each utterance (lens[b]*256); the peak is
 * searched over the utterance's own samples only -- what FastDiff.py:110 sees for a batch of one -- and pcm behind them is 0.
 * valid NULL = the call above.  B <= 4096. */
FD_API int fd_peak_normalize_int16_ragged(fd_handle h, const float *wav, int B, int64_t L, const int64_t *valid, int16_t *pcm,
                                          void *stream);

/* Mel front-end in front of the vocoder: process_utterance(..., vocoder='pwg') of
 * data_gen/tts/data_gen_utils.py:93-147 = librosa.stft(n_fft 1024, hop 256, win 1024, "hann", center, pad_mode "constant") ->
 * magnitude -> librosa.filters.mel(22050, 1024, 80, fmin 80, fmax 7600) -> log10(max(1e-6, .)).
 *   wav [B][n_samples] device, float (int16 PCM / 32768, as librosa.core.load scales it)
 *   mel [B][80][T] device, T <= 1 + n_samples/256 frames (librosa's frame count; the test-time collater then drops the last one).
 * With option "mel" = "tacotron": TacotronSTFT.mel_spectrogram of data_gen/tts/tacotron/layers.py:42-80 (over tacotron/stft.py:78-104,
 * as vocoder_binarizer_tacotron.py:110-116 drives it for FastDiff_tacotron.yaml): the signal reflect-padded by 512 instead of
 * zero-padded (needs n_samples > 512), filters.mel(22050, 1024, 80, 0, 8000), ln(clamp(., 1e-5)). */
FD_API int fd_mel_spectrogram(fd_handle h, const float *wav, int B, int64_t n_samples, float *mel, int T, void *stream);

/* The mel filter bank of the front-end selected by option "mel" -- the matrix the reference gets from
 * librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) at data_gen/tts/data_gen_utils.py:122-134 ('pwg') and
 * data_gen/tts/tacotron/layers.py:42-60 (TacotronSTFT.mel_basis).
 *   fd_set_mel_filterbank: fb [80][513] HOST, row-major (librosa's own layout) -- the weights are then used exactly as given (per filter
 *     the span first..last non-zero bin, summed in ascending bin order); fb NULL restores the default.  A deployment that has librosa
 *     passes `librosa.filters.mel(sr=22050, n_fft=1024, n_mels=80, fmin=80, fmax=7600)` ('pwg') or `(..., fmin=0, fmax=8000)`
 *     (Tacotron) itself.  Takes effect for the calls enqueued after it; n_mels / n_bins must be 80 / 513.
 *   fd_get_mel_filterbank: copies the bank in use to fb_out [80][513] host; returns 1 if it was supplied by the caller, 0 if it is the
 *     default, < 0 on error.
 * The DEFAULT is a restatement of librosa's published algorithm (Slaney scale, area-normalised triangles) -- librosa is absent from the
 * build image; its values are pinned on an independent derivation and on a third-party implementation of librosa.filters.mel
 * (transformers.audio_utils.mel_filter_bank, slaney / slaney: equal to 2e-16; tests/test_mel_frontend.py), not on librosa itself. */
FD_API int fd_set_mel_filterbank(fd_handle h, const float *fb, int n_mels, int n_bins);
FD_API int fd_get_mel_filterbank(fd_handle h, float *fb_out, int n_mels, int n_bins);

/* Test / introspection hooks (not on the reference's API surface) -------------------------------------- */

/* Copies an intermediate of the LAST fd_forward to host (synchronises).  Names: "noise" [B,3,80], "a0".."a3",
 * "kp_h<n>" [B,64,T], "kpack<n>" [B,T,24832] (packed predicted kernels+bias of block n), "x<n>" [B,32,L_n],
 * "range_flags" (32 int32 bit patterns: [0] predictor GEMM, [1 + 4*block + layer] LVC layer, [13 + d] DBlock d,
 * [16 + n] ConvTranspose of block n -- set when an operand of the last fd_forward did not fit fp16 and the fp32 kernel redid
 * that launch; fd_sample clears them every step), "range_flags_call" (the same 32 words OR-ed over all steps of the last fd_sample).
 * Returns the number of floats (also when host_dst is NULL), or a negative status. */
FD_API int64_t fd_read_tap(fd_handle h, const char *name, float *host_dst, int64_t capacity);

/* Position of predicted-kernel element (layer, in, out, tap), and of predicted bias (layer, out), inside one frame's
 * 24832-float packed record.  Lets tests unpack "kpack<n>" into the reference's [B,4,32,64,3,T] / [B,4,64,T] views
 * (modules.py:333-342). */
FD_API int fd_kernel_index(int layer, int in_ch, int out_ch, int tap);
FD_API int fd_bias_index(int layer, int out_ch);

/* Per-kernel timing gathered with hipEvents on the launch stream while option "profile"="1" (graph off).
 * Fills up to `capacity` entries; returns the number of distinct kernels. */
typedef struct fd_kernel_stat {
    char name[48];
    int64_t launches;
    double total_ms;
} fd_kernel_stat;
FD_API int fd_get_profile(fd_handle h, fd_kernel_stat *stats, int capacity);
FD_API int fd_reset_profile(fd_handle h);

/* Bookkeeping of the host-checked range fallback (option "fallback" = "host") and of the graph cache.  Names:
 *   "pieces"         8-step pieces the last fd_sample of more than 8 steps was enqueued as (0 for a shorter call);
 *   "pieces_redone"  of those, the pieces that raised a range flag and were run again from the saved x (settles a pending last piece);
 *   "pieces_fp32"    of those, the pieces enqueued with stages already on their fp32 kernels (after an earlier piece had flagged them);
 *   "fp32_mask"      the flag words (bit i = word i of "range_flags") those later pieces ran on fp32 -- sticky over the call;
 *   "calls_redone"   fd_sample calls of up to 8 steps run again as a whole since fd_create;
 *   "graph_captures" / "graph_hits" / "graph_evictions"   fd_sample's graph look-ups since fd_create that captured a new graph / found
 *                    one / pushed the least recently used one out; "graphs_resident" / "graphs_retired" = kept now / evicted but not
 *                    yet destroyed (their last replay has not completed).
 * Returns the value (>= 0) or a negative status. */
FD_API int64_t fd_get_counter(fd_handle h, const char *name);

#ifdef __cplusplus
}
#endif
#endif /* FASTDIFF_HIP_EXT_H */
