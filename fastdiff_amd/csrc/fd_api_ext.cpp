// fd_api_ext.cpp -- host side of the entry points of include/fastdiff_hip_ext.h: the mel front-end in front of the vocoder and the int16
// waveform epilogue behind it, taps, layout introspection, counters and per-kernel profiling.
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fd_kernels.h"
#include "fd_host.h"

extern "C" {

// The DEFAULT filter bank of a front-end, dense [80][513]: librosa.filters.mel(22050, 1024, 80, fmin, fmax) restated in double
// precision (Slaney mel scale, triangular weights on the FFT bin centres, each filter scaled by 2 / (f[m+2] - f[m])) for 'pwg' (fmin
// 80, fmax 7600; base.yaml:8-9) and Tacotron (0, 8000; FastDiff_tacotron.yaml:20-21).  librosa is not in this image, so these values
// are a restatement pinned only against an independent derivation (tests/test_mel_frontend.py); a deployment that has librosa hands
// its own matrix to fd_set_mel_filterbank and this function is then not used.
static const int MEL_NM = 80, MEL_NB = 513;
static std::vector<float> default_mel_bank(int variant)
{
    const double sr = 22050.0;
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    auto hz_to_mel = [&](double f) { return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp; };
    auto mel_to_hz = [&](double m) { return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m; };
    const double band[MEL_VARIANTS][2] = {{80.0, 7600.0}, {0.0, 8000.0}};
    std::vector<double> mf(MEL_NM + 2);
    const double m0 = hz_to_mel(band[variant][0]), m1 = hz_to_mel(band[variant][1]);
    for (int i = 0; i < MEL_NM + 2; ++i) mf[i] = mel_to_hz(m0 + (m1 - m0) * i / (MEL_NM + 1));
    std::vector<float> fb((size_t)MEL_NM * MEL_NB, 0.0f);
    for (int m = 0; m < MEL_NM; ++m) {
        const double enorm = 2.0 / (mf[m + 2] - mf[m]);
        for (int k = 0; k < MEL_NB; ++k) {
            const double fk = (sr / 2.0) * k / (MEL_NB - 1);
            const double lower = (fk - mf[m]) / (mf[m + 1] - mf[m]), upper = (mf[m + 2] - fk) / (mf[m + 2] - mf[m + 1]);
            const double wv = std::max(0.0, std::min(lower, upper));
            if (wv > 0.0) fb[(size_t)m * MEL_NB + k] = (float)(wv * enorm);
        }
    }
    return fb;
}

static int mel_upload(fd_handle h, const void *src, size_t bytes, const void **dst)
{
    void *d = nullptr;
    FD_HIP(h, hipMalloc(&d, bytes));
    h->mel_allocs.push_back(d);        // freed at fd_destroy: a replaced table stays valid for launches still in flight
    FD_HIP(h, hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    *dst = d;
    return FD_OK;
}

// A dense bank [80][513] as the kernel reads it: per filter the first non-zero bin, the span up to the last non-zero one, and the
// weights of that span exactly as given (k_mel_frontend adds w[k] * |X_k| over the span in ascending k; a zero inside it adds zero).
static int upload_mel_bank(fd_handle h, int variant, const float *fb)
{
    std::vector<int> lo(MEL_NM, 0), cnt(MEL_NM, 0), off(MEL_NM, 0);
    std::vector<float> wts;
    for (int m = 0; m < MEL_NM; ++m) {
        int first = -1, last = -1;
        for (int k = 0; k < MEL_NB; ++k)
            if (fb[(size_t)m * MEL_NB + k] != 0.0f) { if (first < 0) first = k; last = k; }
        off[m] = (int)wts.size();
        if (first >= 0) {
            lo[m] = first; cnt[m] = last - first + 1;
            wts.insert(wts.end(), fb + (size_t)m * MEL_NB + first, fb + (size_t)m * MEL_NB + last + 1);
        }
    }
    if (wts.empty()) wts.push_back(0.0f);
    MelTables t = h->mel[variant];
    int rc;
    if ((rc = mel_upload(h, lo.data(), lo.size() * sizeof(int), reinterpret_cast<const void **>(&t.fb_lo))) != FD_OK) return rc;
    if ((rc = mel_upload(h, cnt.data(), cnt.size() * sizeof(int), reinterpret_cast<const void **>(&t.fb_n))) != FD_OK) return rc;
    if ((rc = mel_upload(h, off.data(), off.size() * sizeof(int), reinterpret_cast<const void **>(&t.fb_off))) != FD_OK) return rc;
    if ((rc = mel_upload(h, wts.data(), wts.size() * sizeof(float), reinterpret_cast<const void **>(&t.fb_w))) != FD_OK) return rc;
    h->mel[variant] = t;
    h->mel_bank[variant].assign(fb, fb + (size_t)MEL_NM * MEL_NB);
    return FD_OK;
}

// Tables of the mel front-end: twiddles and the periodic Hann window in double precision (shared), and each front-end's filter bank
// (the caller's, if fd_set_mel_filterbank supplied one before the first use, else the default above).
static int ensure_mel_tables(fd_handle h)
{
    if (h->mel[MEL_VARIANTS - 1].tab) return FD_OK;
    const int NF = 1024;
    const double pi = 3.14159265358979323846;
    std::vector<float> tab(3 * NF);
    for (int i = 0; i < NF; ++i) {
        tab[i] = (float)cos(2.0 * pi * i / NF);
        tab[NF + i] = (float)sin(2.0 * pi * i / NF);
        tab[2 * NF + i] = (float)(0.5 - 0.5 * cos(2.0 * pi * i / NF));
    }
    const float *tab_dev = nullptr;
    int rc;
    if ((rc = mel_upload(h, tab.data(), tab.size() * sizeof(float), reinterpret_cast<const void **>(&tab_dev))) != FD_OK) return rc;
    for (int v = 0; v < MEL_VARIANTS; ++v) {
        if (!h->mel[v].fb_w) {
            const std::vector<float> fb = default_mel_bank(v);
            if ((rc = upload_mel_bank(h, v, fb.data())) != FD_OK) return rc;
        }
        h->mel[v].tab = tab_dev;                           // last: marks this variant ready
    }
    return FD_OK;
}

int fd_set_mel_filterbank(fd_handle h, const float *fb, int n_mels, int n_bins)
{
    if (!h) return FD_ERR_INVALID;
    if (n_mels != MEL_NM || n_bins != MEL_NB)
        FD_FAIL(h, FD_ERR_INVALID, "fd_set_mel_filterbank: the front-end is 80 filters over the 513 bins of a 1024-point FFT, got [%d][%d]", n_mels, n_bins);
    FD_HIP(h, hipSetDevice(h->device));
    const int v = h->mel_variant;
    if (!fb) {                                             // back to the restated default
        const std::vector<float> def = default_mel_bank(v);
        h->mel_bank_user[v] = false;
        return upload_mel_bank(h, v, def.data());
    }
    for (size_t i = 0; i < (size_t)MEL_NM * MEL_NB; ++i) {      // (on the bit pattern: this file is built with -fno-honor-nans)
        uint32_t bits;
        memcpy(&bits, fb + i, sizeof(bits));
        if ((bits & 0x7F800000u) == 0x7F800000u) FD_FAIL(h, FD_ERR_INVALID, "fd_set_mel_filterbank: element %zu is not finite", i);
    }
    const int rc = upload_mel_bank(h, v, fb);
    if (rc == FD_OK) h->mel_bank_user[v] = true;
    return rc;
}

int fd_get_mel_filterbank(fd_handle h, float *fb_out, int n_mels, int n_bins)
{
    if (!h || !fb_out) return FD_ERR_INVALID;
    if (n_mels != MEL_NM || n_bins != MEL_NB) FD_FAIL(h, FD_ERR_INVALID, "fd_get_mel_filterbank: expects [80][513], got [%d][%d]", n_mels, n_bins);
    FD_HIP(h, hipSetDevice(h->device));
    const int rc = ensure_mel_tables(h);
    if (rc != FD_OK) return rc;
    const int v = h->mel_variant;
    memcpy(fb_out, h->mel_bank[v].data(), sizeof(float) * MEL_NM * MEL_NB);
    return h->mel_bank_user[v] ? 1 : 0;
}

int fd_mel_spectrogram(fd_handle h, const float *wav, int B, int64_t n_samples, float *mel, int T, void *stream)
{
    if (!h || !wav || !mel || B <= 0 || n_samples <= 0 || B > 65535) return FD_ERR_INVALID;
    if (T < 1 || T > 1 + n_samples / 256) FD_FAIL(h, FD_ERR_INVALID, "fd_mel_spectrogram: T=%d outside 1..1+n_samples/256=%lld", T, (long long)(1 + n_samples / 256));
    if (h->mel_variant == MEL_TACOTRON && n_samples <= 512)      // F.pad(mode='reflect') needs pad < length (tacotron/stft.py:84-88)
        FD_FAIL(h, FD_ERR_INVALID, "fd_mel_spectrogram: reflect padding of 512 needs more than 512 samples, got %lld", (long long)n_samples);
    FD_HIP(h, hipSetDevice(h->device));
    int rc = (h->pending.active && h->pending.lazy) ? FD_OK : fd_settle(h);      // the front-end touches no sampler state
    if (rc != FD_OK) return rc;
    rc = ensure_mel_tables(h);
    if (rc != FD_OK) return rc;
    fdk::Launch L = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::mel_frontend(L, wav, B, n_samples, mel, T);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_mel_spectrogram: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_peak_normalize_int16_ragged(fd_handle h, const float *wav, int B, int64_t len, const int64_t *valid, int16_t *pcm, void *stream)
{
    if (!h || !wav || !pcm || B <= 0 || len <= 0 || B > 4096) return FD_ERR_INVALID;
    FD_HIP(h, hipSetDevice(h->device));
    if (!(h->pending.active && h->pending.lazy)) {      // a lazily checked call stays pending: the epilogue's result is provisional with it
        const int rcs = fd_settle(h);
        if (rcs != FD_OK) return rcs;
    }
    const long long *valid_dev = nullptr;
    if (valid) {
        for (int b = 0; b < B; ++b)
            if (valid[b] < 1 || valid[b] > len) FD_FAIL(h, FD_ERR_INVALID, "fd_peak_normalize_int16_ragged: valid[%d] = %lld outside [1, %lld]", b, (long long)valid[b], (long long)len);
        fd_context::StageSlot *sl = nullptr;
        int rc = fd_stage_acquire(h, sizeof(long long) * B, &sl);
        if (rc != FD_OK) return rc;
        for (int b = 0; b < B; ++b) reinterpret_cast<long long *>(sl->host)[b] = valid[b];
        long long *dst = reinterpret_cast<long long *>(reinterpret_cast<char *>(h->scratch) + 32768);      // behind the abs-max words
        FD_HIP(h, hipMemcpyAsync(dst, sl->host, sizeof(long long) * B, hipMemcpyHostToDevice, (hipStream_t)stream));
        if ((rc = fd_stage_commit(h, sl, (hipStream_t)stream)) != FD_OK) return rc;
        valid_dev = dst;
    }
    fdk::Launch L = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::peak_normalize_int16(L, wav, B, len, pcm, valid_dev);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_peak_normalize_int16: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_peak_normalize_int16(fd_handle h, const float *wav, int B, int64_t len, int16_t *pcm, void *stream)
{
    return fd_peak_normalize_int16_ragged(h, wav, B, len, nullptr, pcm, stream);
}

int64_t fd_read_tap(fd_handle h, const char *name, float *host_dst, int64_t capacity)
{
    if (!h || !name) return FD_ERR_INVALID;
    {
        const int rcs = fd_settle(h);
        if (rcs != FD_OK) return rcs;
    }
    if (h->gen) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_read_tap: intermediates are kept by the tuned kernel set only (base.yaml's architecture)");
    const int B = h->last_B, T = h->last_T;
    if (B == 0) FD_FAIL(h, FD_ERR_STATE, "fd_read_tap: no forward has run yet");
    const Workspace &w = h->ws;
    const int64_t L = (int64_t)T * fd::HOPT;
    const std::string k(name);
    const float *src = nullptr;
    int64_t n = 0;
    if (k == "noise") { src = w.noise; n = (int64_t)B * fd::NBLK * fd::COND; }
    else if (k == "a0") { src = w.a[0]; n = B * fd::C * L; }
    else if (k == "a1") { src = w.a[1]; n = B * fd::C * L / 4; }
    else if (k == "a2") { src = w.a[2]; n = B * fd::C * L / 32; }
    else if (k == "a3") { src = w.a[3]; n = (int64_t)B * fd::C * T; }
    else if (k.size() == 6 && k.compare(0, 5, "kpack") == 0 && k[5] >= '0' && k[5] <= '2') {
        n = (int64_t)B * T * fd::KREC; src = w.kpack + (k[5] - '0') * n;
    } else if (k.size() == 5 && k.compare(0, 4, "kp_h") == 0 && k[4] >= '0' && k[4] <= '2') {
        n = (int64_t)B * fd::HID * T; src = w.kp_hB + (k[4] - '0') * n;
    } else if (k.size() == 2 && k[0] == 'x' && k[1] >= '0' && k[1] <= '2') {
        if (!h->keep_taps) FD_FAIL(h, FD_ERR_STATE, "fd_read_tap: set option taps=1 before the forward to keep block outputs");
        const int blk = k[1] - '0';
        src = w.xtap[blk]; n = (int64_t)B * fd::C * T * fd::hop(blk);
    } else if (k == "range_flags") {       // 32 int32 (bit patterns): fp16-range flags of the last step, see Workspace::range_flag
        src = reinterpret_cast<const float *>(w.range_flag); n = 32;
    } else if (k == "range_flags_call") {  // the same, OR-ed over every step since the start of the last call (words 64..95)
        src = reinterpret_cast<const float *>(w.range_flag + 64); n = 32;
    } else FD_FAIL(h, FD_ERR_INVALID, "fd_read_tap: unknown tap '%s'", name);
    if (!host_dst) return n;
    if (capacity < n) FD_FAIL(h, FD_ERR_INVALID, "fd_read_tap: capacity %lld < %lld", (long long)capacity, (long long)n);
    FD_HIP(h, hipSetDevice(h->device));
    FD_HIP(h, hipDeviceSynchronize());
    FD_HIP(h, hipMemcpy(host_dst, src, sizeof(float) * n, hipMemcpyDeviceToHost));
    return n;
}

int fd_kernel_index(int layer, int in_ch, int out_ch, int tap)
{
    if (layer < 0 || layer >= fd::LAYERS || in_ch < 0 || in_ch >= fd::C || out_ch < 0 || out_ch >= 2 * fd::C || tap < 0 || tap >= 3)
        return FD_ERR_INVALID;
    return fd::kernel_index(layer, in_ch, out_ch, tap);
}

int fd_bias_index(int layer, int out_ch)
{
    if (layer < 0 || layer >= fd::LAYERS || out_ch < 0 || out_ch >= 2 * fd::C) return FD_ERR_INVALID;
    return fd::bias_index(layer, out_ch);
}

int64_t fd_get_counter(fd_handle h, const char *name)
{
    if (!h || !name) return FD_ERR_INVALID;
    const std::string k(name);
    if (k == "pieces_redone" || k == "fp32_mask") {      // the last piece of a long call may still be waiting for its check
        const int rcs = fd_settle(h);
        if (rcs != FD_OK) return rcs;
    }
    if (k == "pieces") return h->n_pieces;
    if (k == "pieces_redone") return h->n_pieces_redone;
    if (k == "pieces_fp32") return h->n_pieces_fp32;
    if (k == "fp32_mask") return (int64_t)h->call_fp32_mask;
    if (k == "calls_redone") return h->n_calls_redone;
    if (k == "graph_captures") return h->n_graph_captures;
    if (k == "graph_hits") return h->n_graph_hits;
    if (k == "graph_evictions") return h->n_graph_evictions;
    if (k == "graphs_resident") return (int64_t)h->graphs.size();
    if (k == "graphs_retired") return (int64_t)h->retired.size();
    FD_FAIL(h, FD_ERR_INVALID, "fd_get_counter: unknown counter '%s'", name);
}

int fd_get_profile(fd_handle h, fd_kernel_stat *stats, int capacity)
{
    if (!h) return FD_ERR_INVALID;
    hipSetDevice(h->device);
    fd_prof_drain(h);
    int i = 0;
    for (const auto &kv : h->prof_acc) {
        if (stats && i < capacity) {
            memset(&stats[i], 0, sizeof(fd_kernel_stat));
            strncpy(stats[i].name, kv.first.c_str(), sizeof(stats[i].name) - 1);
            stats[i].launches = kv.second.first;
            stats[i].total_ms = kv.second.second;
        }
        ++i;
    }
    return i;
}

int fd_reset_profile(fd_handle h)
{
    if (!h) return FD_ERR_INVALID;
    fd_prof_drain(h);
    h->prof_acc.clear();
    return FD_OK;
}

}  // extern "C"
