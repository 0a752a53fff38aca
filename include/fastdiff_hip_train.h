/*
 * fastdiff_hip_train.h -- libfastdiff_hip.so: the training-side operators: the denoiser's layers as
 * differentiable forward / backward pairs in the reference's own tensor layouts, so that theta_timestep_loss
 * (modules/FastDiff/module/util.py:291-325) runs on HIP kernels under PyTorch autograd (fastdiff_amd/train.py, lvc_op.py).
 * Not part of the inference boundary.  Conventions: fastdiff_hip.h; every call is asynchronous on `stream`.
 */
#ifndef FASTDIFF_HIP_TRAIN_H
#define FASTDIFF_HIP_TRAIN_H

#include "fastdiff_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Training side: TimeAware_LVCBlock.location_variable_convolution (modules/FastDiff/module/modules.py:220-253,
 * dilation = 1 as at its only call site, modules.py:216) as a differentiable operator in the reference's own tensor layouts, so that
 * theta_timestep_loss (util.py:291-325) can differentiate through it while the rest of the module stays on PyTorch autograd:
 *   out[b,o,q] = bias[b,o,q/hop] + sum_{i,k} xpad[b,i,q+k-(ks-1)/2] * kernel[b,i,o,k,q/hop]
 *   x [B,Cin,T*hop]   kernel [B,Cin,Cout,ks,T]   bias [B,Cout,T]   out, dout [B,Cout,T*hop]      (all device, float32, contiguous)
 * fd_lvc_backward writes the gradients whose pointer is not NULL: dx (needs kernel), dkernel and dbias (need x).
 * Any handle of the device will do (it supplies the device, the error text and, for the model's own shape -- Cin 32, Cout 64, ks 3,
 * hop 8 / 64 / 256, which runs on the fp32 matrix instruction -- a scratch buffer of B*T*Cin*Cout*ks floats for the frame-major copy of
 * the kernels: calls on one handle must therefore be ordered on one stream, as autograd orders a forward and its backward);
 * Cin*Cout*ks <= 8192, ks odd. */
FD_API int fd_lvc_forward(fd_handle h, const float *x, const float *kernel, const float *bias, int B, int Cin, int Cout, int ks, int T,
                          int hop, float *out, void *stream);
FD_API int fd_lvc_backward(fd_handle h, const float *x, const float *kernel, const float *dout, int B, int Cin, int Cout, int ks, int T,
                           int hop, float *dx, float *dkernel, float *dbias, void *stream);
/* The same with a batch stride on the predicted kernel and on its gradient (floats between two utterances; 0 = Cin*Cout*ks*T, a tensor
 * of its own): one layer's slice [:, i] of the predictor's [B, layers, Cin, Cout, ks, T] output -- and of the gradient buffer of that
 * shape -- is used where it lies, without a contiguous copy each way (the model's shape only: 32 -> 64, k3, hop 8 / 64 / 256). */
FD_API int fd_lvc_forward_strided(fd_handle h, const float *x, const float *kernel, int64_t kernel_bstride, const float *bias, int B, int Cin,
                                  int Cout, int ks, int T, int hop, float *out, void *stream);
FD_API int fd_lvc_backward_strided(fd_handle h, const float *x, const float *kernel, int64_t kernel_bstride, const float *dout, int B, int Cin,
                                   int Cout, int ks, int T, int hop, float *dx, float *dkernel, int64_t dkernel_bstride, float *dbias,
                                   void *stream);

/* KernelPredictor.kernel_conv (modules/FastDiff/module/modules.py:315-318,330-331: Conv1d(64 -> M, kernel 3, padding 1) with
 * M = lvc_layers * in * 2 in * 3 = 24576) for the training path, in the reference's layouts: x [B,64,T], weight [M,64,3] (after
 * weight-norm), bias [M], out / dout [B,M,T] (device, float32, contiguous); the three gradients whose pointer is not NULL are written
 * (dx needs weight; dweight and dbias need x).  fp32 matrix instruction throughout.  M a multiple of 128, 1 <= T <= 128 (the
 * reference trains on crops of 100 frames: base.yaml:50-51) -- anything else returns FD_ERR_UNSUPPORTED and the caller keeps its own
 * convolution. */
FD_API int fd_kconv_forward(fd_handle h, const float *x, const float *weight, const float *bias, int B, int M, int T, float *out, void *stream);
FD_API int fd_kconv_backward(fd_handle h, const float *x, const float *weight, const float *dout, int B, int M, int T, float *dx,
                             float *dweight, float *dbias, void *stream);
/* The same with the activation the predictor puts behind its small convolutions (modules.py:296-314: Conv1d, LeakyReLU(0.1)) inside:
 * out = leaky_relu(conv, post_slope); the backward takes that output (y) and dout = the gradient behind the activation.  M <= 512
 * (input and residual convolutions: M = 64); post_slope = 1 is the plain convolution (y may then be NULL).
 * in_slope (a chain of such pairs, e.g. the six of the predictor's residual stack, where x is itself the activated output of the
 * pair below and has no other reader): dx comes out multiplied by THAT activation's mask (x > 0 ? 1 : in_slope), i.e. as the
 * gradient in front of it, and the pair below is then called with post_slope = 1 on that gradient; 1 = dx as it is. */
FD_API int fd_kconv_forward_act(fd_handle h, const float *x, const float *weight, const float *bias, int B, int M, int T, float post_slope,
                                float *out, void *stream);
FD_API int fd_kconv_backward_act(fd_handle h, const float *x, const float *weight, const float *y, const float *dout, int B, int M, int T,
                                 float post_slope, float in_slope, float *dx, float *dweight, float *dbias, void *stream);
/* The weight and bias gradients of n <= 8 such convolutions of ONE shape in two launches: x, dout, y, dweight, dbias are HOST arrays of n
 * device pointers (y[i] = NULL: dout[i] is already the gradient in front of the activation; y = NULL: none is masked).  For the six pairs
 * of the predictor's residual stack once its dx chain (fd_kconv_backward_act with dweight = dbias = NULL) has run: one launch of
 * 6 x B workgroups instead of six latency-bound launches of B. */
FD_API int fd_kconv_backward_w_multi(fd_handle h, int n, const float *const *x, const float *const *dout, const float *const *y, int B, int M,
                                     int T, float post_slope, float *const *dweight, float *const *dbias, void *stream);

/* A skip tensor's fan-out on the training path (FastDiff_model.py:91-98): x [rows = B*C, L] is read by the DiffusionDBlock below it, which
 * begins by picking every factor-th column (F.interpolate to L / factor, nearest: modules.py:128-131), and as `audio_down` by the four
 * layers of the LVC block at its rate (modules.py:209).  fd_fan_forward: picked [rows, L / factor] = x[:, ::factor].  fd_fan_backward:
 * dx = g0 + g1 + g2 + g3 + scatter(gpicked) in one pass (any of the five may be NULL = no gradient from that reader); under autograd
 * the same is a zero-fill, a strided scatter and four full-size additions.  L a multiple of factor. */
FD_API int fd_fan_forward(fd_handle h, const float *x, int rows, int64_t L, int factor, float *picked, void *stream);
FD_API int fd_fan_backward(fd_handle h, const float *g0, const float *g1, const float *g2, const float *g3, const float *gpicked, int rows,
                           int64_t L, int factor, float *dx, void *stream);

/* KernelPredictor.input_conv (modules.py:292-295: Conv1d(80 -> 64, kernel 5, padding 2), LeakyReLU(0.1)) for the training path as one
 * operator each way: x [B,80,T], weight [64,80,5], bias [64], out / y / dout [B,64,T] (device, float32, contiguous), 1 <= T <= 128;
 * out = leaky_relu(conv, post_slope); the backward takes that output (y) and dout = the gradient behind the activation, and writes the
 * gradients whose pointer is not NULL (dweight / dbias: per-utterance partial sums added in a fixed order). */
FD_API int fd_input_conv_forward(fd_handle h, const float *x, const float *weight, const float *bias, int B, int T, float post_slope, float *out,
                                 void *stream);
FD_API int fd_input_conv_backward(fd_handle h, const float *x, const float *weight, const float *y, const float *dout, int B, int T,
                                  float post_slope, float *dx, float *dweight, float *dbias, void *stream);

/* Side by side: n <= 8 INDEPENDENT convolutions of one shape in one launch each.  The network's three KernelPredictors have identical
 * front ends -- input convolution, then six Conv1d(64, 64, 3) + LeakyReLU pairs -- on different weights and inputs; each is a chain
 * of latency-bound launches of B workgroups, the three together the same chain with 3 B.  Every pointer argument is a HOST array of n
 * device pointers (the library passes them on as kernel arguments); shapes and meaning per item as in the one-convolution entry
 * points above.  fd_kconv_backward_x_multi is one step of n dx chains (dx only: the weight gradients come from
 * fd_kconv_backward_w_multi once the chains have run); y[i] / dweight[i] / dbias[i] may be NULL where the single entry point allows it. */
FD_API int fd_kconv_forward_act_multi(fd_handle h, int n, const float *const *x, const float *const *weight, const float *const *bias, int B,
                                      int M, int T, float post_slope, float *const *out, void *stream);
FD_API int fd_kconv_backward_x_multi(fd_handle h, int n, const float *const *x, const float *const *weight, const float *const *y,
                                     const float *const *dout, int B, int M, int T, float post_slope, float in_slope, float *const *dx,
                                     void *stream);
FD_API int fd_input_conv_forward_multi(fd_handle h, int n, const float *const *x, const float *const *weight, const float *const *bias, int B,
                                       int T, float post_slope, float *const *out, void *stream);
FD_API int fd_input_conv_backward_multi(fd_handle h, int n, const float *const *x, const float *const *weight, const float *const *y,
                                        const float *const *dout, int B, int T, float post_slope, float *const *dx, float *const *dweight,
                                        float *const *dbias, void *stream);

/* The same two operators joined without the reference's tensor in between ("frames").  The reference hands the predicted kernels from
 * kernel_conv to the location-variable convolution as [B, layers, 32, 64, 3, T] (modules.py:333-338; T innermost), which the matrix
 * kernels of the operator have to transpose into frame-major order before use (and the gradient back): three passes over 6144*B*T
 * floats per layer and training step that exist only because of that layout.  Here kernel_conv writes
 *     frames [B, layers, T, 6144]      (M = layers * 6144; one frame = the operator's forward operand order)
 * and reads the gradient in the same shape (one frame = the operator's dK accumulator order), and the operator takes one layer's
 * [T, 6144] block per utterance where it lies: kernel_frames / dkernel_frames point at utterance 0's block of the layer, *_bstride =
 * floats between two utterances (layers * T * 6144); bias / dbias [64, T] per utterance likewise take the floats between two utterances
 * (0 = 64 * T; layers * 64 * T for one layer's slice of bias_conv's [B, layers, 64, T] output).  Both orders are permutations of the 6144 coefficients of a frame, internal to
 * this library (csrc/fd_frame_order.h); fastdiff_amd.lvc_op.frames_to_reference / reference_to_frames convert for inspection.  Same
 * shapes and limits as above (operator: 32 -> 64 channels, k 3, hop 8 / 64 / 256; kernel_conv: M a multiple of 6144, 1 <= T <= 128);
 * results equal those of the entry points above bit for bit (same products, same summation order), except kernel_conv's dx, whose sum
 * over the M rows runs frame group by frame group instead of row by row (float32 rounding apart). */
FD_API int fd_kconv_forward_frames(fd_handle h, const float *x, const float *weight, const float *bias, int B, int M, int T, float *frames,
                                   void *stream);
FD_API int fd_kconv_backward_frames(fd_handle h, const float *x, const float *weight, const float *dframes, int B, int M, int T, float *dx,
                                    float *dweight, float *dbias, void *stream);
FD_API int fd_lvc_forward_frames(fd_handle h, const float *x, const float *kernel_frames, int64_t kernel_bstride, const float *bias,
                                 int64_t bias_bstride, int B, int T, int hop, float *out, void *stream);
FD_API int fd_lvc_backward_frames(fd_handle h, const float *x, const float *kernel_frames, int64_t kernel_bstride, const float *dout, int B,
                                  int T, int hop, float *dx, float *dkernel_frames, int64_t dkernel_bstride, float *dbias,
                                  int64_t dbias_bstride, void *stream);

/* The gate of an LVC layer with its residual (modules.py:217) for the training path: out = x + sigmoid(y[:, :C]) * tanh(y[:, C:]),
 * x, out, dout [B,C,L], y, dy [B,2C,L] (device, float32, contiguous).  Under autograd the reference runs twelve elementwise kernels
 * for this line (four forward, eight backward), each moving the layer's whole tensor through HBM; these are one pass each way.
 * d out / d x is the identity, so fd_gate_backward only produces dy. */
FD_API int fd_gate_forward(fd_handle h, const float *x, const float *y, int B, int C, int64_t L, float *out, void *stream);
FD_API int fd_gate_backward(fd_handle h, const float *y, const float *dout, int B, int C, int64_t L, float *dy, void *stream);

/* The two 7-tap convolutions at the ends of the network on the training path: which = 0 first_audio_conv = Conv1d(1, 32, 7, padding 3)
 * (FastDiff_model.py:34-36,89): x [B,1,L] -> y [B,32,L], weight [32,1,7]; which = 1 final_conv = Conv1d(32, 1, 7, padding 3)
 * (FastDiff_model.py:67-68,100): x [B,32,L] -> y [B,1,L], weight [1,32,7].  L a multiple of 4.  backward writes dx (nullable),
 * dweight and dbias (each nullable) from x, the folded weight and dy; sums in a fixed order. */
FD_API int fd_conv7_forward(fd_handle h, int which, const float *x, const float *weight, const float *bias, int B, int64_t L, float *y,
                            void *stream);
FD_API int fd_conv7_backward(fd_handle h, int which, const float *x, const float *weight, const float *dy, int B, int64_t L, float *dx,
                             float *dweight, float *dbias, void *stream);

/* The block's up-sampler on the training path: `self.upsample(F.leaky_relu(x, 0.2))`, upsample = ConvTranspose1d(32, 32, 2 r, stride r,
 * padding r / 2) (modules/FastDiff/module/modules.py:163-166,205-206), ratio r = 4 or 8:  x [B,32,Lin] -> y [B,32,Lin*r]; weight
 * [32 in, 32 out, 2 r] (torch's ConvTranspose1d layout, no weight-norm), bias [32].  backward: from x, weight, dy it writes dx (the
 * activation's mask applied), dweight, dbias (each nullable); sums in a fixed order. */
FD_API int fd_upsample_forward(fd_handle h, const float *x, const float *weight, const float *bias, int B, int64_t Lin, int ratio, float *y,
                               void *stream);
FD_API int fd_upsample_backward(fd_handle h, const float *x, const float *weight, const float *dy, int B, int64_t Lin, int ratio, float *dx,
                                float *dweight, float *dbias, void *stream);

/* Weight-norm of the training path: every Conv1d of the model carries torch.nn.utils.weight_norm (FastDiff_model.py:71-72,115-122), i.e.
 * its forward evaluates w = torch._weight_norm(v, g, 0): w[r, :] = v[r, :] * g[r] / ||v[r, :]|| on the [rows = out channels, cols = in * k]
 * view.  forward also leaves ||v[r]|| in norm [rows] for the backward, which turns dw into dv [rows, cols] and dg [rows]. */
FD_API int fd_weight_norm_forward(fd_handle h, const float *v, const float *g, int64_t rows, int cols, float *w, float *norm, void *stream);
FD_API int fd_weight_norm_backward(fd_handle h, const float *v, const float *g, const float *norm, const float *dw, int64_t rows, int cols,
                                   float *dv, float *dg, void *stream);
/* The same for n parameter tensors in ceil(n / 28) launches each way (the model has 53 weight-normed convolutions: 106 launches of a
 * few microseconds per training step otherwise).  items: n records in HOST memory -- the library passes them on as kernel arguments, so
 * nothing is uploaded and a captured graph depends on no table's lifetime; every pointer inside is a device pointer: forward reads
 * v, g and writes w, norm; backward reads v, g, norm, dw and writes dv, dg (dw == NULL: that weight took no part in the loss, its dv and
 * dg are zeroed). */
typedef struct fd_wn_item {
    const float *v, *g;      /* [rows, cols], [rows] */
    float *w, *norm;         /* [rows, cols], [rows] */
    const float *dw;         /* [rows, cols] or NULL */
    float *dv, *dg;          /* [rows, cols], [rows] */
    int64_t rows;
    int32_t cols, reserved;
} fd_wn_item;
FD_API int fd_weight_norm_multi_forward(fd_handle h, const fd_wn_item *items, int n, void *stream);
FD_API int fd_weight_norm_multi_backward(fd_handle h, const fd_wn_item *items, int n, void *stream);

/* The denoiser's 21 small convolutions on the training path -- DiffusionDBlock.conv[0..2] applied as `layer(F.leaky_relu(x, 0.2))`
 * (modules/FastDiff/module/modules.py:120-125,136-137) and TimeAware_LVCBlock.convs[0..3] applied as `x += audio_down;
 * y = F.leaky_relu(conv(F.leaky_relu(x, 0.2)), 0.2)` (modules.py:183-187,209-212) -- as one differentiable operator:
 *   xs = x (+ skip);   y = post(bias + conv1d(pre(xs), weight, dilation, padding = dilation)),   pre / post = leaky_relu with the given
 *   slope, slope 1 = no activation.   x, skip, xs, y, dy, gxs, dxs [B,32,L];  weight [32,32,3] (folded: the caller applies weight-norm),
 *   bias [32];  L a multiple of 4, dilation one of 1, 2, 3, 4, 9, 27.
 * forward: skip may be NULL (then xs = x and xs_out may be NULL); with a skip xs_out receives x + skip (the layer's gate reads it).
 * backward: xs = the convolution's un-activated input (x + skip, or x), y = the forward's output (its sign is the post-activation's
 * mask), gxs (nullable) = the gradient that reached xs from its other readers; writes dxs = gxs + pre'(xs) * (W^T * (dy * post'(y)))
 * (the gradient of x and of skip alike), dweight [32,32,3], dbias [32] (each nullable).  Sums are formed in a fixed order. */
FD_API int fd_conv32_forward(fd_handle h, const float *x, const float *skip, const float *weight, const float *bias, int B, int64_t L,
                             int dilation, float pre_slope, float post_slope, float *xs_out, float *y, void *stream);
FD_API int fd_conv32_backward(fd_handle h, const float *xs, const float *y, const float *weight, const float *dy, const float *gxs, int B,
                              int64_t L, int dilation, float pre_slope, float post_slope, float *dxs, float *dweight, float *dbias,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTDIFF_HIP_TRAIN_H */
