// fd_api_train.cpp -- host side of the training operators of include/fastdiff_hip_train.h: argument checks, scratch
// buffers and launches of fd_kernels_train / _kconv / _cconv.hip.  Nothing here is on the inference path.
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fd_kernels.h"
#include "fd_host.h"

extern "C" {

static int check_lvc_op(fd_handle h, int B, int Cin, int Cout, int ks, int T, int hop, const char *who)
{
    if (B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || hop <= 0 || ks <= 0 || (ks & 1) == 0)
        FD_FAIL(h, FD_ERR_INVALID, "%s: B=%d Cin=%d Cout=%d ks=%d T=%d hop=%d must be positive, ks odd", who, B, Cin, Cout, ks, T, hop);
    if ((int64_t)Cin * Cout * ks > 8192 || Cout > 256)
        FD_FAIL(h, FD_ERR_UNSUPPORTED, "%s: Cin*Cout*ks = %lld > 8192 (or Cout > 256) has no kernel", who, (long long)Cin * Cout * ks);
    if ((int64_t)B * std::max(Cin, Cout) * T * hop >= (int64_t)1 << 31)
        FD_FAIL(h, FD_ERR_INVALID, "%s: tensor too large for one call", who);
    if (B > 65535 || std::max(Cin, Cout) > 65535) FD_FAIL(h, FD_ERR_INVALID, "%s: B, channels <= 65535", who);
    return FD_OK;
}

// The matrix-pipe kernels of the operator read the predicted kernels frame-major: room for one copy (B*T*Cin*Cout*ks floats), kept on
// the handle and grown when a call needs more (hipFree waits for the device, so work in flight on the old buffer is safe).  Calls on
// one handle share it: they must be ordered on one stream, as torch.autograd orders a forward and its backward.
static int lvc_scratch(fd_handle h, int B, int Cin, int Cout, int ks, int T, int hop, float **out)
{
    *out = nullptr;
    if (!fdk::lvc_op_needs_scratch(Cin, Cout, ks, hop)) return FD_OK;
    const size_t bytes = sizeof(float) * (size_t)B * T * Cin * Cout * ks;
    if (h->lvc_scratch_bytes < bytes) {
        if (h->lvc_scratch) FD_HIP(h, hipFree(h->lvc_scratch));
        h->lvc_scratch = nullptr; h->lvc_scratch_bytes = 0;
        FD_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->lvc_scratch), bytes));
        h->lvc_scratch_bytes = bytes;
    }
    *out = h->lvc_scratch;
    return FD_OK;
}

int fd_lvc_forward_strided(fd_handle h, const float *x, const float *kernel, int64_t kernel_bstride, const float *bias, int B, int Cin, int Cout,
                           int ks, int T, int hop, float *out, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !kernel || !bias || !out) FD_FAIL(h, FD_ERR_INVALID, "fd_lvc_forward: null pointer");
    int rc = check_lvc_op(h, B, Cin, Cout, ks, T, hop, "fd_lvc_forward");
    if (rc != FD_OK) return rc;
    const int64_t own = (int64_t)Cin * Cout * ks * T;
    if (kernel_bstride != 0 && kernel_bstride != own && (kernel_bstride < own || !fdk::lvc_op_needs_scratch(Cin, Cout, ks, hop)))
        FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_lvc_forward: a batch-strided kernel (stride %lld) needs the model's shape (32 -> 64, k3, hop 8 / 64 / 256)", (long long)kernel_bstride);
    FD_HIP(h, hipSetDevice(h->device));
    float *scratch = nullptr;
    if ((rc = lvc_scratch(h, B, Cin, Cout, ks, T, hop, &scratch)) != FD_OK) return rc;
    fdk::Launch L = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::lvc_op_forward(L, x, kernel, bias, out, B, Cin, Cout, ks, T, hop, scratch, kernel_bstride);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_lvc_forward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_lvc_forward(fd_handle h, const float *x, const float *kernel, const float *bias, int B, int Cin, int Cout, int ks, int T, int hop,
                   float *out, void *stream)
{
    return fd_lvc_forward_strided(h, x, kernel, 0, bias, B, Cin, Cout, ks, T, hop, out, stream);
}

int fd_lvc_backward_strided(fd_handle h, const float *x, const float *kernel, int64_t kernel_bstride, const float *dout, int B, int Cin, int Cout,
                            int ks, int T, int hop, float *dx, float *dkernel, int64_t dkernel_bstride, float *dbias, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!dout || ((dkernel || dbias) && !x) || (dx && !kernel)) FD_FAIL(h, FD_ERR_INVALID, "fd_lvc_backward: null pointer");
    int rc = check_lvc_op(h, B, Cin, Cout, ks, T, hop, "fd_lvc_backward");
    if (rc != FD_OK) return rc;
    const int64_t own = (int64_t)Cin * Cout * ks * T;
    for (int64_t st : {kernel_bstride, dkernel_bstride})
        if (st != 0 && st != own && (st < own || !fdk::lvc_op_needs_scratch(Cin, Cout, ks, hop)))
            FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_lvc_backward: a batch-strided kernel / dkernel (stride %lld) needs the model's shape (32 -> 64, k3, hop 8 / 64 / 256)", (long long)st);
    FD_HIP(h, hipSetDevice(h->device));
    float *scratch = nullptr;
    if ((rc = lvc_scratch(h, B, Cin, Cout, ks, T, hop, &scratch)) != FD_OK) return rc;
    if (scratch && dx && !kernel) FD_FAIL(h, FD_ERR_INVALID, "fd_lvc_backward: null pointer");
    fdk::Launch L = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::lvc_op_backward(L, x, kernel, dout, dx, dkernel, dbias, B, Cin, Cout, ks, T, hop, scratch, kernel_bstride, dkernel_bstride);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_lvc_backward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_lvc_backward(fd_handle h, const float *x, const float *kernel, const float *dout, int B, int Cin, int Cout, int ks, int T, int hop,
                    float *dx, float *dkernel, float *dbias, void *stream)
{
    return fd_lvc_backward_strided(h, x, kernel, 0, dout, B, Cin, Cout, ks, T, hop, dx, dkernel, 0, dbias, stream);
}

// kernel_conv of the KernelPredictor (training path).  The backward adds up partial sums (row slices for dx, utterance ranges for
// dweight / dbias, each in a fixed order) through a scratch buffer kept on the handle next to the LVC operator's (same rule: calls on
// one handle are ordered on one stream).
static int kconv_scratch_reserve(fd_handle h, int B, int M, int T)
{
    const size_t bytes = sizeof(float) * fdk::kconv_scratch_floats(B, M, T);
    if (h->kconv_scratch_bytes < bytes) {
        if (h->kconv_scratch) FD_HIP(h, hipFree(h->kconv_scratch));
        h->kconv_scratch = nullptr; h->kconv_scratch_bytes = 0;
        FD_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->kconv_scratch), bytes));
        h->kconv_scratch_bytes = bytes;
    }
    return FD_OK;
}

static int check_act(fd_handle h, int M, int T, float post, const char *who)
{
    if (!(post > 0.0f && post <= 1.0f)) FD_FAIL(h, FD_ERR_INVALID, "%s: the leaky-relu slope must lie in (0, 1] (1 = no activation), got %g", who, post);
    if (post != 1.0f && !fdk::kconv_act_supported(M, T)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "%s: the fused activation covers M <= 512 (the predictor's small convolutions), got M=%d", who, M);
    return FD_OK;
}

int fd_kconv_forward_act(fd_handle h, const float *x, const float *weight, const float *bias, int B, int M, int T, float post_slope, float *out,
                         void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !weight || !bias || !out) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_forward: null pointer");
    if (B <= 0 || B > 65535) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_forward: B=%d", B);
    if (!fdk::kconv_supported(M, T)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_kconv_forward: M=%d (a multiple of 32) and T=%d (1..128) only", M, T);
    const int rc = check_act(h, M, T, post_slope, "fd_kconv_forward");
    if (rc != FD_OK) return rc;
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::kconv_forward(La, x, weight, bias, out, B, M, T, false, post_slope);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_kconv_forward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_kconv_forward(fd_handle h, const float *x, const float *weight, const float *bias, int B, int M, int T, float *out, void *stream)
{
    return fd_kconv_forward_act(h, x, weight, bias, B, M, T, 1.0f, out, stream);
}

int fd_kconv_backward_act(fd_handle h, const float *x, const float *weight, const float *y, const float *dout, int B, int M, int T,
                          float post_slope, float in_slope, float *dx, float *dweight, float *dbias, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!dout || ((dweight || dbias) && !x) || (dx && !weight)) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward: null pointer");
    if (B <= 0 || B > 65535) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward: B=%d", B);
    if (!fdk::kconv_supported(M, T)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_kconv_backward: M=%d (a multiple of 32) and T=%d (1..128) only", M, T);
    int rc = check_act(h, M, T, post_slope, "fd_kconv_backward");
    if (rc != FD_OK) return rc;
    if (post_slope != 1.0f && !y) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward: a fused activation needs the forward's output y");
    if (!(in_slope > 0.0f && in_slope <= 1.0f)) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward: in_slope must lie in (0, 1], got %g", in_slope);
    if (in_slope != 1.0f && dx && !x) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward: in_slope needs x");
    FD_HIP(h, hipSetDevice(h->device));
    if ((rc = kconv_scratch_reserve(h, B, M, T)) != FD_OK) return rc;
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::kconv_backward(La, x, weight, dout, dx, dweight, dbias, B, M, T, h->kconv_scratch, false, post_slope != 1.0f ? y : nullptr, post_slope,
                                       in_slope);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_kconv_backward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_kconv_backward_w_multi(fd_handle h, int n, const float *const *x, const float *const *dout, const float *const *y, int B, int M, int T,
                              float post_slope, float *const *dweight, float *const *dbias, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !dout || (!dweight && !dbias)) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward_w_multi: null pointer");
    if (n < 1 || n > 8) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward_w_multi: n=%d outside 1..8", n);
    if (B <= 0 || B > 65535) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward_w_multi: B=%d", B);
    if (!fdk::kconv_act_supported(M, T)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_kconv_backward_w_multi: M=%d (a multiple of 32, <= 512) and T=%d (1..128) only", M, T);
    int rc = check_act(h, M, T, post_slope, "fd_kconv_backward_w_multi");
    if (rc != FD_OK) return rc;
    for (int i = 0; i < n; ++i)
        if (!x[i] || !dout[i]) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward_w_multi: null pointer in item %d", i);
    FD_HIP(h, hipSetDevice(h->device));
    {
        const size_t bytes = sizeof(float) * fdk::kconv_w_multi_scratch_floats(n, B, M);
        if (h->kconv_scratch_bytes < bytes) {
            if (h->kconv_scratch) FD_HIP(h, hipFree(h->kconv_scratch));
            h->kconv_scratch = nullptr; h->kconv_scratch_bytes = 0;
            FD_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->kconv_scratch), bytes));
            h->kconv_scratch_bytes = bytes;
        }
    }
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::kconv_backward_w_multi(La, n, x, dout, y, post_slope, B, M, T, dweight, dbias, h->kconv_scratch);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_kconv_backward_w_multi: %s", hipGetErrorString(e));
    return FD_OK;
}

static int check_input_conv(fd_handle h, int B, int T, float post, const char *who);

static int kconv_scratch_floats_reserve(fd_handle h, size_t floats)
{
    const size_t bytes = sizeof(float) * floats;
    if (h->kconv_scratch_bytes < bytes) {
        if (h->kconv_scratch) FD_HIP(h, hipFree(h->kconv_scratch));
        h->kconv_scratch = nullptr; h->kconv_scratch_bytes = 0;
        FD_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->kconv_scratch), bytes));
        h->kconv_scratch_bytes = bytes;
    }
    return FD_OK;
}

static int check_multi(fd_handle h, int n, int B, const void *const *lists, int nlists, const char *who)
{
    if (n < 1 || n > 8) FD_FAIL(h, FD_ERR_INVALID, "%s: n=%d outside 1..8", who, n);
    if (B <= 0 || B > 65535) FD_FAIL(h, FD_ERR_INVALID, "%s: B=%d", who, B);
    for (int k = 0; k < nlists; ++k) {
        const void *const *l = reinterpret_cast<const void *const *>(lists[k]);
        if (!l) FD_FAIL(h, FD_ERR_INVALID, "%s: null pointer list", who);
        for (int i = 0; i < n; ++i)
            if (!l[i]) FD_FAIL(h, FD_ERR_INVALID, "%s: null pointer in item %d", who, i);
    }
    return FD_OK;
}

int fd_kconv_forward_act_multi(fd_handle h, int n, const float *const *x, const float *const *weight, const float *const *bias, int B, int M, int T,
                               float post_slope, float *const *out, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    const void *lists[4] = {x, weight, bias, out};
    int rc = check_multi(h, n, B, lists, 4, "fd_kconv_forward_act_multi");
    if (rc != FD_OK) return rc;
    if (!fdk::kconv_act_supported(M, T)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_kconv_forward_act_multi: M=%d (a multiple of 32, <= 512) and T=%d (1..128) only", M, T);
    if ((rc = check_act(h, M, T, post_slope, "fd_kconv_forward_act_multi")) != FD_OK) return rc;
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::kconv_forward_multi(La, n, x, weight, bias, out, B, M, T, post_slope);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_kconv_forward_act_multi: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_kconv_backward_x_multi(fd_handle h, int n, const float *const *x, const float *const *weight, const float *const *y, const float *const *dout,
                              int B, int M, int T, float post_slope, float in_slope, float *const *dx, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    const void *lists[3] = {weight, dout, dx};
    int rc = check_multi(h, n, B, lists, 3, "fd_kconv_backward_x_multi");
    if (rc != FD_OK) return rc;
    if (!fdk::kconv_act_supported(M, T)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_kconv_backward_x_multi: M=%d (a multiple of 32, <= 512) and T=%d (1..128) only", M, T);
    if ((rc = check_act(h, M, T, post_slope, "fd_kconv_backward_x_multi")) != FD_OK) return rc;
    if (!(in_slope > 0.0f && in_slope <= 1.0f)) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward_x_multi: in_slope must lie in (0, 1], got %g", in_slope);
    if (in_slope != 1.0f) {
        const void *lx[1] = {x};
        if ((rc = check_multi(h, n, B, lx, 1, "fd_kconv_backward_x_multi")) != FD_OK) return rc;
    }
    if (post_slope != 1.0f && !y) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward_x_multi: a fused activation needs the forward's outputs y");
    FD_HIP(h, hipSetDevice(h->device));
    if ((rc = kconv_scratch_floats_reserve(h, fdk::kconv_x_multi_scratch_floats(n, B, M, T))) != FD_OK) return rc;
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::kconv_backward_x_multi(La, n, x, weight, post_slope != 1.0f ? y : nullptr, dout, dx, B, M, T, post_slope, in_slope, h->kconv_scratch);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_kconv_backward_x_multi: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_input_conv_forward_multi(fd_handle h, int n, const float *const *x, const float *const *weight, const float *const *bias, int B, int T,
                                float post_slope, float *const *out, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    const void *lists[4] = {x, weight, bias, out};
    int rc = check_multi(h, n, B, lists, 4, "fd_input_conv_forward_multi");
    if (rc != FD_OK) return rc;
    if ((rc = check_input_conv(h, B, T, post_slope, "fd_input_conv_forward_multi")) != FD_OK) return rc;
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::input_conv_forward_multi(La, n, x, weight, bias, out, B, T, post_slope);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_input_conv_forward_multi: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_input_conv_backward_multi(fd_handle h, int n, const float *const *x, const float *const *weight, const float *const *y, const float *const *dout,
                                 int B, int T, float post_slope, float *const *dx, float *const *dweight, float *const *dbias, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    const void *lists[4] = {x, weight, y, dout};
    int rc = check_multi(h, n, B, lists, 4, "fd_input_conv_backward_multi");
    if (rc != FD_OK) return rc;
    if ((rc = check_input_conv(h, B, T, post_slope, "fd_input_conv_backward_multi")) != FD_OK) return rc;
    if (dx) { const void *l[1] = {dx}; if ((rc = check_multi(h, n, B, l, 1, "fd_input_conv_backward_multi")) != FD_OK) return rc; }
    FD_HIP(h, hipSetDevice(h->device));
    if ((rc = kconv_scratch_floats_reserve(h, fdk::input_conv_multi_scratch_floats(n, B))) != FD_OK) return rc;
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::input_conv_backward_multi(La, n, x, weight, y, dout, dx, dweight, dbias, B, T, post_slope, h->kconv_scratch);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_input_conv_backward_multi: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_kconv_backward(fd_handle h, const float *x, const float *weight, const float *dout, int B, int M, int T, float *dx, float *dweight,
                      float *dbias, void *stream)
{
    return fd_kconv_backward_act(h, x, weight, nullptr, dout, B, M, T, 1.0f, 1.0f, dx, dweight, dbias, stream);
}

static int weight_norm_multi(fd_handle h, const fd_wn_item *items, int n, void *stream, bool backward, const char *who)
{
    if (!h) return FD_ERR_INVALID;
    if (!items) FD_FAIL(h, FD_ERR_INVALID, "%s: null pointer", who);
    if (n <= 0 || n > 4096) FD_FAIL(h, FD_ERR_INVALID, "%s: n=%d", who, n);
    for (int i = 0; i < n; ++i) {
        const fd_wn_item &I = items[i];
        if (I.rows <= 0 || I.cols <= 0 || I.rows > ((int64_t)1 << 31) || !I.v || !I.g || !I.norm || (backward ? (!I.dv || !I.dg) : !I.w))
            FD_FAIL(h, FD_ERR_INVALID, "%s: item %d: rows=%lld cols=%d or a null pointer", who, i, (long long)I.rows, I.cols);
    }
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::weight_norm_multi(La, items, n, backward);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    return FD_OK;
}
int fd_weight_norm_multi_forward(fd_handle h, const fd_wn_item *items, int n, void *stream)
{
    return weight_norm_multi(h, items, n, stream, false, "fd_weight_norm_multi_forward");
}
int fd_weight_norm_multi_backward(fd_handle h, const fd_wn_item *items, int n, void *stream)
{
    return weight_norm_multi(h, items, n, stream, true, "fd_weight_norm_multi_backward");
}

// A skip tensor's fan-out (fd_kernels_train.hip: k_fan_*).
int fd_fan_forward(fd_handle h, const float *x, int rows, int64_t L, int factor, float *picked, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !picked) FD_FAIL(h, FD_ERR_INVALID, "fd_fan_forward: null pointer");
    if (rows <= 0 || rows > 65535 || L <= 0 || factor < 1 || L % factor != 0) FD_FAIL(h, FD_ERR_INVALID, "fd_fan_forward: rows=%d L=%lld factor=%d", rows, (long long)L, factor);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::fan_pick(La, x, picked, rows, L, factor);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_fan_forward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_fan_backward(fd_handle h, const float *g0, const float *g1, const float *g2, const float *g3, const float *gpicked, int rows, int64_t L,
                    int factor, float *dx, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!dx) FD_FAIL(h, FD_ERR_INVALID, "fd_fan_backward: null pointer");
    if (rows <= 0 || rows > 65535 || L <= 0 || factor < 1 || L % factor != 0) FD_FAIL(h, FD_ERR_INVALID, "fd_fan_backward: rows=%d L=%lld factor=%d", rows, (long long)L, factor);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    const float *g[4] = {g0, g1, g2, g3};
    hipError_t e = fdk::fan_sum(La, g, gpicked, dx, rows, L, factor);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_fan_backward: %s", hipGetErrorString(e));
    return FD_OK;
}

// The predictor's input convolution (80 -> 64, k5) with its activation (fd_kernels_kconv.hip: k_ic_*); per-utterance partial sums of
// the weight gradient in the kernel_conv scratch.
static int check_input_conv(fd_handle h, int B, int T, float post, const char *who)
{
    if (B <= 0 || B > 65535) FD_FAIL(h, FD_ERR_INVALID, "%s: B=%d", who, B);
    if (T < 1 || T > 128) FD_FAIL(h, FD_ERR_UNSUPPORTED, "%s: T=%d (1..128) only", who, T);
    if (!(post > 0.0f && post <= 1.0f)) FD_FAIL(h, FD_ERR_INVALID, "%s: the leaky-relu slope must lie in (0, 1] (1 = no activation), got %g", who, post);
    return FD_OK;
}

int fd_input_conv_forward(fd_handle h, const float *x, const float *weight, const float *bias, int B, int T, float post_slope, float *out, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !weight || !bias || !out) FD_FAIL(h, FD_ERR_INVALID, "fd_input_conv_forward: null pointer");
    const int rc = check_input_conv(h, B, T, post_slope, "fd_input_conv_forward");
    if (rc != FD_OK) return rc;
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::input_conv_forward(La, x, weight, bias, out, B, T, post_slope);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_input_conv_forward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_input_conv_backward(fd_handle h, const float *x, const float *weight, const float *y, const float *dout, int B, int T, float post_slope,
                           float *dx, float *dweight, float *dbias, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!dout || !y || ((dweight || dbias) && !x) || (dx && !weight)) FD_FAIL(h, FD_ERR_INVALID, "fd_input_conv_backward: null pointer");
    int rc = check_input_conv(h, B, T, post_slope, "fd_input_conv_backward");
    if (rc != FD_OK) return rc;
    FD_HIP(h, hipSetDevice(h->device));
    {
        const size_t bytes = sizeof(float) * fdk::input_conv_scratch_floats(B);
        if (h->kconv_scratch_bytes < bytes) {
            if (h->kconv_scratch) FD_HIP(h, hipFree(h->kconv_scratch));
            h->kconv_scratch = nullptr; h->kconv_scratch_bytes = 0;
            FD_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->kconv_scratch), bytes));
            h->kconv_scratch_bytes = bytes;
        }
    }
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::input_conv_backward(La, x, weight, y, dout, dx, dweight, dbias, B, T, post_slope, h->kconv_scratch);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_input_conv_backward: %s", hipGetErrorString(e));
    return FD_OK;
}

// "frames": kernel_conv and the operator joined through frame-major tensors (include/fastdiff_hip.h)

int fd_kconv_forward_frames(fd_handle h, const float *x, const float *weight, const float *bias, int B, int M, int T, float *frames, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !weight || !bias || !frames) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_forward_frames: null pointer");
    if (B <= 0 || B > 65535) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_forward_frames: B=%d", B);
    if (!fdk::kconv_frames_supported(M, T)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_kconv_forward_frames: M=%d (a multiple of 6144) and T=%d (1..128) only", M, T);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::kconv_forward(La, x, weight, bias, frames, B, M, T, true);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_kconv_forward_frames: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_kconv_backward_frames(fd_handle h, const float *x, const float *weight, const float *dframes, int B, int M, int T, float *dx,
                             float *dweight, float *dbias, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!dframes || ((dweight || dbias) && !x) || (dx && !weight)) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward_frames: null pointer");
    if (B <= 0 || B > 65535) FD_FAIL(h, FD_ERR_INVALID, "fd_kconv_backward_frames: B=%d", B);
    if (!fdk::kconv_frames_supported(M, T)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_kconv_backward_frames: M=%d (a multiple of 6144) and T=%d (1..128) only", M, T);
    FD_HIP(h, hipSetDevice(h->device));
    const int rc = kconv_scratch_reserve(h, B, M, T);
    if (rc != FD_OK) return rc;
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::kconv_backward(La, x, weight, dframes, dx, dweight, dbias, B, M, T, h->kconv_scratch, true);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_kconv_backward_frames: %s", hipGetErrorString(e));
    return FD_OK;
}

static int check_frames_stride(fd_handle h, int64_t st, int T, const char *who)
{
    if (st < (int64_t)T * 6144 || st % 4 != 0) FD_FAIL(h, FD_ERR_INVALID, "%s: a frame stride of %lld floats (at least T * 6144, a multiple of 4)", who, (long long)st);
    return FD_OK;
}

int fd_lvc_forward_frames(fd_handle h, const float *x, const float *kernel_frames, int64_t kernel_bstride, const float *bias, int64_t bias_bstride,
                          int B, int T, int hop, float *out, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !kernel_frames || !bias || !out) FD_FAIL(h, FD_ERR_INVALID, "fd_lvc_forward_frames: null pointer");
    int rc = check_lvc_op(h, B, 32, 64, 3, T, hop, "fd_lvc_forward_frames");
    if (rc != FD_OK) return rc;
    if (!fdk::lvc_op_needs_scratch(32, 64, 3, hop)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_lvc_forward_frames: hop 8 / 64 / 256 only, got %d", hop);
    if ((rc = check_frames_stride(h, kernel_bstride, T, "fd_lvc_forward_frames")) != FD_OK) return rc;
    if (bias_bstride != 0 && bias_bstride < (int64_t)64 * T) FD_FAIL(h, FD_ERR_INVALID, "fd_lvc_forward_frames: bias stride %lld < 64 * T", (long long)bias_bstride);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch L = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::lvc_op_forward(L, x, kernel_frames, bias, out, B, 32, 64, 3, T, hop, nullptr, kernel_bstride, true, bias_bstride);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_lvc_forward_frames: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_lvc_backward_frames(fd_handle h, const float *x, const float *kernel_frames, int64_t kernel_bstride, const float *dout, int B, int T,
                           int hop, float *dx, float *dkernel_frames, int64_t dkernel_bstride, float *dbias, int64_t dbias_bstride, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!dout || ((dkernel_frames || dbias) && !x) || (dx && !kernel_frames)) FD_FAIL(h, FD_ERR_INVALID, "fd_lvc_backward_frames: null pointer");
    int rc = check_lvc_op(h, B, 32, 64, 3, T, hop, "fd_lvc_backward_frames");
    if (rc != FD_OK) return rc;
    if (!fdk::lvc_op_needs_scratch(32, 64, 3, hop)) FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_lvc_backward_frames: hop 8 / 64 / 256 only, got %d", hop);
    if (dx && (rc = check_frames_stride(h, kernel_bstride, T, "fd_lvc_backward_frames")) != FD_OK) return rc;
    if (dkernel_frames && (rc = check_frames_stride(h, dkernel_bstride, T, "fd_lvc_backward_frames")) != FD_OK) return rc;
    if (dbias && dbias_bstride != 0 && dbias_bstride < (int64_t)64 * T) FD_FAIL(h, FD_ERR_INVALID, "fd_lvc_backward_frames: dbias stride %lld < 64 * T", (long long)dbias_bstride);
    FD_HIP(h, hipSetDevice(h->device));
    float *scratch = nullptr;
    if (dx && !h->lvc_dx_gather && (rc = lvc_scratch(h, B, 32, 64, 3, T, hop, &scratch)) != FD_OK) return rc;      // option lvc_dx = copy: the dx kernel's operand order
    fdk::Launch L = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::lvc_op_backward(L, x, kernel_frames, dout, dx, dkernel_frames, dbias, B, 32, 64, 3, T, hop, scratch, kernel_bstride,
                                        dkernel_bstride, true, dbias_bstride);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_lvc_backward_frames: %s", hipGetErrorString(e));
    return FD_OK;
}

static int cconv_scratch_reserve(fd_handle h, size_t floats)
{
    const size_t bytes = sizeof(float) * floats;
    if (h->cconv_scratch_bytes < bytes) {
        if (h->cconv_scratch) FD_HIP(h, hipFree(h->cconv_scratch));
        h->cconv_scratch = nullptr; h->cconv_scratch_bytes = 0;
        FD_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->cconv_scratch), bytes));
        h->cconv_scratch_bytes = bytes;
    }
    return FD_OK;
}

// The small convolutions of the training path (fd_kernels_cconv.hip).  The backward's per-workgroup partial sums live in a scratch
// buffer on the handle (calls on one handle are ordered on one stream, as for the operators above).
static int check_conv32(fd_handle h, int B, int64_t L, int dil, float pre, float post, const char *who)
{
    if (B <= 0 || B > 65535 || L <= 0) FD_FAIL(h, FD_ERR_INVALID, "%s: B=%d L=%lld", who, B, (long long)L);
    if (!(pre > 0.0f && pre <= 1.0f) || !(post > 0.0f && post <= 1.0f))
        FD_FAIL(h, FD_ERR_INVALID, "%s: leaky-relu slopes must lie in (0, 1] (1 = no activation), got %g / %g", who, pre, post);
    if (!fdk::cconv_supported(dil, L))
        FD_FAIL(h, FD_ERR_UNSUPPORTED, "%s: dilation %d (1, 2, 3, 4, 9, 27) and a length that is a multiple of 4 only, got L=%lld", who, dil, (long long)L);
    if ((int64_t)B * 32 * L >= (int64_t)1 << 40) FD_FAIL(h, FD_ERR_INVALID, "%s: tensor too large", who);
    return FD_OK;
}

int fd_conv32_forward(fd_handle h, const float *x, const float *skip, const float *weight, const float *bias, int B, int64_t L, int dilation,
                      float pre_slope, float post_slope, float *xs_out, float *y, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !weight || !bias || !y || (skip && !xs_out)) FD_FAIL(h, FD_ERR_INVALID, "fd_conv32_forward: null pointer (xs_out is needed with a skip)");
    int rc = check_conv32(h, B, L, dilation, pre_slope, post_slope, "fd_conv32_forward");
    if (rc != FD_OK) return rc;
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::cconv_forward(La, x, skip, weight, bias, xs_out, y, B, L, dilation, pre_slope, post_slope);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_conv32_forward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_conv32_backward(fd_handle h, const float *xs, const float *y, const float *weight, const float *dy, const float *gxs, int B, int64_t L,
                       int dilation, float pre_slope, float post_slope, float *dxs, float *dweight, float *dbias, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!xs || !y || !weight || !dy) FD_FAIL(h, FD_ERR_INVALID, "fd_conv32_backward: null pointer");
    int rc = check_conv32(h, B, L, dilation, pre_slope, post_slope, "fd_conv32_backward");
    if (rc != FD_OK) return rc;
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    rc = cconv_scratch_reserve(h, fdk::cconv_scratch_floats(La, dilation, B, L));
    if (rc != FD_OK) return rc;
    hipError_t e = fdk::cconv_backward(La, xs, y, weight, dy, gxs, dxs, dweight, dbias, B, L, dilation, pre_slope, post_slope, h->cconv_scratch);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_conv32_backward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_conv7_forward(fd_handle h, int which, const float *x, const float *weight, const float *bias, int B, int64_t L, float *y, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !weight || !bias || !y) FD_FAIL(h, FD_ERR_INVALID, "fd_conv7_forward: null pointer");
    if ((which != 0 && which != 1) || B <= 0 || B > 65535 || L < 4 || L % 4 != 0 || L >= ((int64_t)1 << 25))
        FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_conv7_forward: which=%d (0 first_audio_conv, 1 final_conv), B=%d, L=%lld (a multiple of 4)", which, B, (long long)L);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::conv7_forward(La, which, x, weight, bias, y, B, L);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_conv7_forward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_conv7_backward(fd_handle h, int which, const float *x, const float *weight, const float *dy, int B, int64_t L, float *dx, float *dweight,
                      float *dbias, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !weight || !dy) FD_FAIL(h, FD_ERR_INVALID, "fd_conv7_backward: null pointer");
    if ((which != 0 && which != 1) || B <= 0 || B > 65535 || L < 4 || L % 4 != 0 || L >= ((int64_t)1 << 25))
        FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_conv7_backward: which=%d (0 first_audio_conv, 1 final_conv), B=%d, L=%lld (a multiple of 4)", which, B, (long long)L);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    int rc = cconv_scratch_reserve(h, fdk::conv7_scratch_floats(La, B, L));
    if (rc != FD_OK) return rc;
    hipError_t e = fdk::conv7_backward(La, which, x, weight, dy, dx, dweight, dbias, B, L, h->cconv_scratch);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_conv7_backward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_upsample_forward(fd_handle h, const float *x, const float *weight, const float *bias, int B, int64_t Lin, int ratio, float *y, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !weight || !bias || !y) FD_FAIL(h, FD_ERR_INVALID, "fd_upsample_forward: null pointer");
    if ((ratio != 4 && ratio != 8) || B <= 0 || B > 65535 || Lin < 1 || Lin * ratio >= ((int64_t)1 << 25))
        FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_upsample_forward: ratio %d (4 or 8), B=%d, Lin=%lld", ratio, B, (long long)Lin);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::convt_forward(La, x, weight, bias, y, B, Lin, ratio);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_upsample_forward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_upsample_backward(fd_handle h, const float *x, const float *weight, const float *dy, int B, int64_t Lin, int ratio, float *dx, float *dweight,
                         float *dbias, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !weight || !dy) FD_FAIL(h, FD_ERR_INVALID, "fd_upsample_backward: null pointer");
    if ((ratio != 4 && ratio != 8) || B <= 0 || B > 65535 || Lin < 1 || Lin * ratio >= ((int64_t)1 << 25))
        FD_FAIL(h, FD_ERR_UNSUPPORTED, "fd_upsample_backward: ratio %d (4 or 8), B=%d, Lin=%lld", ratio, B, (long long)Lin);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    int rc = cconv_scratch_reserve(h, fdk::convt_scratch_floats(La, ratio, B, Lin));
    if (rc != FD_OK) return rc;
    hipError_t e = fdk::convt_backward(La, x, weight, dy, dx, dweight, dbias, B, Lin, ratio, h->cconv_scratch);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_upsample_backward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_weight_norm_forward(fd_handle h, const float *v, const float *g, int64_t rows, int cols, float *w, float *norm, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!v || !g || !w || !norm) FD_FAIL(h, FD_ERR_INVALID, "fd_weight_norm_forward: null pointer");
    if (rows <= 0 || cols <= 0 || rows > ((int64_t)1 << 31)) FD_FAIL(h, FD_ERR_INVALID, "fd_weight_norm_forward: rows=%lld cols=%d", (long long)rows, cols);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::weight_norm_forward(La, v, g, w, norm, rows, cols);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_weight_norm_forward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_weight_norm_backward(fd_handle h, const float *v, const float *g, const float *norm, const float *dw, int64_t rows, int cols, float *dv,
                            float *dg, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!v || !g || !norm || !dw || !dv || !dg) FD_FAIL(h, FD_ERR_INVALID, "fd_weight_norm_backward: null pointer");
    if (rows <= 0 || cols <= 0 || rows > ((int64_t)1 << 31)) FD_FAIL(h, FD_ERR_INVALID, "fd_weight_norm_backward: rows=%lld cols=%d", (long long)rows, cols);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::weight_norm_backward(La, v, g, norm, dw, dv, dg, rows, cols);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_weight_norm_backward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_gate_forward(fd_handle h, const float *x, const float *y, int B, int C, int64_t L, float *out, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!x || !y || !out) FD_FAIL(h, FD_ERR_INVALID, "fd_gate_forward: null pointer");
    if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || C > 65535) FD_FAIL(h, FD_ERR_INVALID, "fd_gate_forward: B=%d C=%d L=%lld", B, C, (long long)L);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::gate_forward(La, x, y, out, B, C, L);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_gate_forward: %s", hipGetErrorString(e));
    return FD_OK;
}

int fd_gate_backward(fd_handle h, const float *y, const float *dout, int B, int C, int64_t L, float *dy, void *stream)
{
    if (!h) return FD_ERR_INVALID;
    if (!y || !dout || !dy) FD_FAIL(h, FD_ERR_INVALID, "fd_gate_backward: null pointer");
    if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || C > 65535) FD_FAIL(h, FD_ERR_INVALID, "fd_gate_backward: B=%d C=%d L=%lld", B, C, (long long)L);
    FD_HIP(h, hipSetDevice(h->device));
    fdk::Launch La = {h, (hipStream_t)stream, false};
    hipError_t e = fdk::gate_backward(La, y, dout, dy, B, C, L);
    if (e != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "fd_gate_backward: %s", hipGetErrorString(e));
    return FD_OK;
}

}  // extern "C"
