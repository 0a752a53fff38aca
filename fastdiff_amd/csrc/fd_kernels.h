// fd_kernels.h -- internal: per-stage entry points of the naive and the fast (MFMA) kernel sets.
#pragma once
#include "fd_internal.h"

namespace fdk {
// naive set (fd_kernels_naive.hip)
hipError_t naive_first_conv(const Launch &L, const StepIO &io, int B, int T);
hipError_t naive_dblock(const Launch &L, int d, int B, int T);
hipError_t naive_kp_front(const Launch &L, const StepIO &io, int B, int T);
hipError_t naive_kp_gemm(const Launch &L, int B, int T);
hipError_t naive_convt(const Launch &L, int n, const float *x_in, float *x_out, int B, int Lin);
hipError_t naive_lvc_layer(const Launch &L, int n, int layer, float *x, const float *skip, float *y, int B, int T);
hipError_t naive_final_eps(const Launch &L, const float *x32, float *eps, int B, int T);
hipError_t naive_update(const Launch &L, float *x, const float *eps, int64_t n);
// fast set (fd_kernels_first_final / _dblock / _kp / _convt / _lvc .hip)
hipError_t fast_first_conv(const Launch &L, const StepIO &io, int B, int T);
hipError_t fast_dblock(const Launch &L, int d, int B, int T, const float *audio);
hipError_t fast_kp_front(const Launch &L, const StepIO &io, int B, int T);
hipError_t fast_kp_gemm(const Launch &L, int B, int T);
hipError_t fast_convt(const Launch &L, int n, const float *x_in, float *x_out, int B, int Lin);
// up: layer 0 of block n >= 1 with the block's ConvTranspose inside it -- x_in is then the block's input (fp16x2-only launches)
hipError_t fast_lvc_layer(const Launch &L, int n, int layer, const float *x_in, const float *skip, float *x_out, int B, int T, bool up = false);
hipError_t fast_final(const Launch &L, const StepIO &io, const float *x32, int B, int T);
// the LVC operator with its gradients (fd_kernels_train.hip)
// scratch: B*T*Cin*Cout*ks floats when lvc_op_needs_scratch (the model's shape: matrix-pipe kernels), else unused
bool lvc_op_needs_scratch(int Cin, int Cout, int ks, int hop);
// kbs / dkbs: floats between two utterances of K / dK (0 = a tensor of its own; the model's shape also takes one layer's slice of a
// [B, layers, Cin, Cout, ks, T] tensor)
// frames (the model's shape): K / dK are frame-major ([T][6144] per utterance, fd_frame_order.h: K in ORDER_FWD, dK in ORDER_DK) as
// kconv_forward / kconv_backward with frames = true write / read them; kbs / dkbs are then the floats between two utterances of those
hipError_t lvc_op_forward(const Launch &L, const float *x, const float *K, const float *bias, float *out, int B, int Cin, int Cout, int ks,
                          int T, int hop, float *scratch, int64_t kbs = 0, bool frames = false, int64_t bbs = 0);
hipError_t lvc_op_backward(const Launch &L, const float *x, const float *K, const float *dout, float *dx, float *dK, float *dbias, int B,
                           int Cin, int Cout, int ks, int T, int hop, float *scratch, int64_t kbs = 0, int64_t dkbs = 0, bool frames = false,
                           int64_t dbbs = 0);      // bbs / dbbs (frames only): floats between two utterances of bias / dbias (0 = Cout * T)
// the gate + residual of an LVC layer, one pass forward and one backward (modules.py:217)
hipError_t gate_forward(const Launch &L, const float *x, const float *y, float *out, int B, int C, int64_t len);
hipError_t gate_backward(const Launch &L, const float *y, const float *dout, float *dy, int B, int C, int64_t len);
// a skip tensor's fan-out on the training path (fd_kernels_train.hip): out[r, j] = x[r, j f] (the DBlock's nearest pick), and the sum of
// the gradients that come back: dx = g[0] + g[1] + g[2] + g[3] + scatter(gp) (null pointers = absent); rows = B * C
hipError_t fan_pick(const Launch &L, const float *x, float *out, int rows, int64_t len, int f);
hipError_t fan_sum(const Launch &L, const float *const g[4], const float *gp, float *dx, int rows, int64_t len, int f);
// the KernelPredictor's kernel_conv (Conv1d 64 -> M, k3) forward and backward for the training path (fd_kernels_kconv.hip);
// scratch: kconv_scratch_floats(B, M, T) floats for the backward's partial sums
bool kconv_supported(int M, int T);
size_t kconv_scratch_floats(int B, int M, int T);
// frames (M a multiple of 6144 = M / 6144 layers of the LVC operator's kernels): out is [B][layers][T][6144] with every frame in the
// operator's forward operand order, dout the same shape in its dK accumulator order (fd_frame_order.h) -- the LVC kernels then read /
// write the predictor's tensors where they lie
bool kconv_frames_supported(int M, int T);
// post / y (M <= 512: the predictor's small convolutions): out = leaky_relu(conv, post); the backward is given that output (y) and
// takes dout as the gradient behind the activation
bool kconv_act_supported(int M, int T);
hipError_t kconv_forward(const Launch &L, const float *h, const float *W, const float *bias, float *out, int B, int M, int T, bool frames = false,
                         float post = 1.0f);
hipError_t kconv_backward(const Launch &L, const float *h, const float *W, const float *dout, float *dh, float *dW, float *dbias, int B, int M,
                          int T, float *scratch, bool frames = false, const float *y = nullptr, float post = 1.0f, float in_slope = 1.0f);
// the weight / bias gradients of n <= 8 small convolutions of ONE shape (M <= 512) in two launches (the six pairs of the predictor's residual
// stack, once its dx chain has run); y[i] != null: dout[i] is masked with that activated output; scratch: kconv_w_multi_scratch_floats()
size_t kconv_w_multi_scratch_floats(int n, int B, int M);
hipError_t kconv_backward_w_multi(const Launch &L, int n, const float *const *h, const float *const *dout, const float *const *y, float post, int B,
                                  int M, int T, float *const *dW, float *const *dbias, float *scratch);
// n <= 8 independent convolutions of one shape side by side in one launch each (the three KernelPredictors' front ends): forward of small
// convolutions, one step of their dx chains (kconv_backward's dh part), the input convolution both ways.  Host arrays of device pointers.
hipError_t kconv_forward_multi(const Launch &L, int n, const float *const *h, const float *const *W, const float *const *bias, float *const *out, int B,
                               int M, int T, float post);
size_t kconv_x_multi_scratch_floats(int n, int B, int M, int T);
hipError_t kconv_backward_x_multi(const Launch &L, int n, const float *const *h, const float *const *W, const float *const *y, const float *const *dout,
                                  float *const *dh, int B, int M, int T, float post, float in_slope, float *scratch);
hipError_t input_conv_forward_multi(const Launch &L, int n, const float *const *x, const float *const *w, const float *const *bias, float *const *out,
                                    int B, int T, float post);
size_t input_conv_multi_scratch_floats(int n, int B);
hipError_t input_conv_backward_multi(const Launch &L, int n, const float *const *x, const float *const *w, const float *const *y, const float *const *dy,
                                     float *const *dx, float *const *dw, float *const *db, int B, int T, float post, float *scratch);
// in_slope != 1 (a chain of such pairs): h is the activated output of the pair below and dh comes out multiplied by that activation's
// mask (h > 0 ? 1 : in_slope), i.e. as the gradient in front of it
// the predictor's input convolution with its activation: leaky_relu(Conv1d(80 -> 64, k5, pad 2), post) (modules.py:292-295), T <= 128;
// the backward takes the activated output y; scratch: input_conv_scratch_floats(B) floats
size_t input_conv_scratch_floats(int B);
hipError_t input_conv_forward(const Launch &L, const float *x, const float *w, const float *bias, float *out, int B, int T, float post);
hipError_t input_conv_backward(const Launch &L, const float *x, const float *w, const float *y, const float *dy, float *dx, float *dw, float *db, int B,
                               int T, float post, float *scratch);
// one layer's "x (+ skip) -> leaky_relu -> dilated Conv1d(32 -> 32, k3) -> bias -> (leaky_relu)" forward and backward for the training
// path (fd_kernels_cconv.hip); scratch: cconv_scratch_floats() floats for the per-workgroup partial sums of dW / db
bool cconv_supported(int dil, int64_t len);
size_t cconv_scratch_floats(const Launch &L, int dil, int B, int64_t len);
hipError_t cconv_forward(const Launch &L, const float *x, const float *skip, const float *w, const float *bias, float *xs_out, float *y, int B,
                         int64_t len, int dil, float pre, float post);
hipError_t cconv_backward(const Launch &L, const float *xs, const float *y, const float *w, const float *dy, const float *gxs, float *dxs,
                          float *dw, float *db, int B, int64_t len, int dil, float pre, float post, float *scratch);
// first_audio_conv (which = 0: Conv1d 1 -> 32, k7) and final_conv (which = 1: Conv1d 32 -> 1, k7) of the training path
size_t conv7_scratch_floats(const Launch &L, int B, int64_t len);
hipError_t conv7_forward(const Launch &L, int which, const float *x, const float *w, const float *bias, float *y, int B, int64_t len);
hipError_t conv7_backward(const Launch &L, int which, const float *x, const float *w, const float *dy, float *dx, float *dw, float *db, int B,
                          int64_t len, float *scratch);
// the block's up-sampler on the training path: leaky_relu(x, 0.2) -> ConvTranspose1d(32, 32, 2 r, stride r, padding r / 2), r = 4 or 8
size_t convt_scratch_floats(const Launch &L, int r, int B, int64_t len_in);
hipError_t convt_forward(const Launch &L, const float *x, const float *w, const float *bias, float *y, int B, int64_t len_in, int r);
hipError_t convt_backward(const Launch &L, const float *x, const float *w, const float *dy, float *dx, float *dw, float *db, int B, int64_t len_in,
                          int r, float *scratch);
// torch._weight_norm(v, g, 0) on a [rows, cols] view and its backward (fd_kernels_cconv.hip)
hipError_t weight_norm_forward(const Launch &L, const float *v, const float *g, float *w, float *norm, int64_t rows, int cols);
hipError_t weight_norm_backward(const Launch &L, const float *v, const float *g, const float *norm, const float *dw, float *dv, float *dg,
                                int64_t rows, int cols);
// ... for n tensors in ceil(n / 28) launches: items in HOST memory (include/fastdiff_hip.h: fd_wn_item), passed on as kernel arguments
hipError_t weight_norm_multi(const Launch &L, const fd_wn_item *items, int n, bool backward);
}  // namespace fdk
