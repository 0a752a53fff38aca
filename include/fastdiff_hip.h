/*
 * fastdiff_hip.h -- C ABI of libfastdiff_hip.so: the MI355X (gfx950) FastDiff vocoder inference path.
 *
 * This is the drop-in boundary under the reference's Python API.  Every entry point names the
 * reference interface it replaces (path:line in Rongjiehuang/FastDiff).  Plain pointers and sizes
 * only; no torch types.  All tensors are float32, contiguous, [batch][channel][time] with time
 * innermost -- the reference's own layout.
 *
 * Conventions
 *   - every function returns 0 on success, a negative fd_status on failure; fd_last_error() gives text.
 *   - a handle is bound to one device and is NOT thread safe (one handle per GPU/stream, like one
 *     reference process per GPU under mp.spawn, utils/trainer.py:94-107).  Its calls share one workspace and are ordered by the
 *     stream they run on: when fd_forward / fd_sample arrive on another stream than the previous call, a pending range check is
 *     settled first and the new stream is made to wait for the tail of the previous call (an event recorded at the end of every
 *     call), so consecutive calls may change streams but never overlap.  A stream may be destroyed once the calls made on it are
 *     SETTLED (fd_sample_check / fd_sample_settle, or any later fd_sample / fd_forward on the handle has returned): a pending
 *     range check may have to run its call again on that stream; the library never touches a previous call's stream otherwise.
 *   - device pointers are caller-owned; all work is enqueued asynchronously on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream).  No hidden synchronisation except
 *     in fd_create / fd_destroy / fd_commit_weights / fd_read_tap / workspace growth.
 *   - there is no CPU fallback: without a usable HIP device every compute call fails with FD_ERR_HIP.
 */
#ifndef FASTDIFF_HIP_H
#define FASTDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_API __attribute__((visibility("default")))

typedef struct fd_context *fd_handle;

enum fd_status {
    FD_OK = 0,
    FD_ERR_INVALID = -1,      /* bad argument (the reference would raise AssertionError / ValueError)     */
    FD_ERR_UNSUPPORTED = -2,  /* architecture hyper-parameters this build has no kernels for                */
    FD_ERR_HIP = -3,          /* HIP runtime error (no device, launch failure, out of memory ...)            */
    FD_ERR_STATE = -4,        /* call order (e.g. forward before fd_commit_weights)                          */
    FD_ERR_MISSING = -5       /* a state_dict tensor was never provided                                      */
};

/* Constructor arguments of FastDiff(...) -- modules/FastDiff/module/FastDiff_model.py:13-26;
 * filled from hparams at modules/FastDiff/task/FastDiff.py:17-29 (modules/FastDiff/config/base.yaml:21-33). */
typedef struct fd_config {
    int audio_channels;                 /* 1   */
    int inner_channels;                 /* 32  */
    int cond_channels;                  /* 80  */
    int n_upsample;                     /* 3   */
    int upsample_ratios[8];             /* 8,8,4 */
    int lvc_layers_each_block;          /* 4   */
    int lvc_kernel_size;                /* 3   */
    int kpnet_hidden_channels;          /* 64  */
    int kpnet_conv_size;                /* 3   */
    int diffusion_step_embed_dim_in;    /* 128 */
    int diffusion_step_embed_dim_mid;   /* 512 */
    int diffusion_step_embed_dim_out;   /* 512 */
    int use_weight_norm;                /* 1   */
} fd_config;

/* One reverse step of sampling_given_noise_schedule (modules/FastDiff/module/util.py:216-229).
 * The host (Python shim) derives these with the reference's own fp32 arithmetic (util.py:187-204):
 *   t      = steps_infer[n]                       mapped (fractional) diffusion step fed to the net (:217)
 *   c_eps  = beta_n / sqrt(1 - alpha_hat_n^2)     x -= c_eps * eps                                   (:226)
 *   c_div  = sqrt(1 - beta_n)                     x /= c_div                                         (:227)
 *   sigma  = sigma_hat_n                          x  = x + sigma * z   when add_noise (n > 0)        (:228-229)
 *   c1,c2,c3                                      the "ddim" branch x = c1*x + c2*eps + c3*eps       (:219-224)
 * table[0] is the step executed FIRST (n = N-1), table[N-1] the last (n = 0). */
typedef struct fd_step {
    float t, c_eps, c_div, sigma, c1, c2, c3;
    int32_t add_noise;
} fd_step;

/* Fills *cfg with the reference defaults above. */
FD_API int fd_default_config(fd_config *cfg);

/* FastDiff(**cfg).cuda(device): creates kernels' context on HIP device `device`.
 * Replaces: FastDiff.__init__ + .cuda() (FastDiff_model.py:13-72; FastDiff.py:17-29,  egs/demo.ipynb cell 0).
 * Any configuration the reference constructor accepts is taken.  base.yaml's architecture (the defaults above: every shipped YAML) runs
 * on the tuned gfx950 kernel set; any other one (other channel counts, ratios incl. odd ones, 1..8 LVC layers, odd kernel sizes, other
 * embedding widths) on runtime-shaped exact-fp32 kernels (csrc/fd_generic.hip: a correctness path -- no graph, no
 * fd_read_tap).  FD_ERR_UNSUPPORTED only for what the reference's own forward / sampler cannot run: audio_channels != 1 (first_audio_conv
 * is Conv1d(1, C), FastDiff_model.py:34), even lvc_kernel_size / kpnet_conv_size (the sequence length changes, modules.py:183-187,293-318),
 * odd diffusion_step_embed_dim_in (util.py:423). */
FD_API int fd_create(const fd_config *cfg, int device, fd_handle *out);
FD_API int fd_destroy(fd_handle h);

/* Text of the last error on this handle (h may be NULL: last error of a failed fd_create). */
FD_API const char *fd_last_error(fd_handle h);

/* load_state_dict(ckpt['state_dict']['model']) -- utils/trainer.py:355-356; egs/demo.ipynb cell 0.
 * `name` is the reference state_dict key ("lvc_blocks.0.kernel_predictor.kernel_conv.weight_v", "fc_t1.bias" ...;
 * after remove_weight_norm() the "*.weight" form is accepted instead of weight_g/weight_v, FastDiff_model.py:104-113).
 * `host_data` is a HOST pointer, copied before return.  dims must equal the reference shape. */
FD_API int fd_set_weight(fd_handle h, const char *name, const float *host_data, const int64_t *dims, int ndim);

/* Folds weight-norm (w = g*v/||v||, FastDiff_model.py:115-122), repacks for the kernels and uploads.
 * Fails with FD_ERR_MISSING naming the first absent tensor. */
FD_API int fd_commit_weights(fd_handle h);

/* eps = FastDiff.forward((x, mel, steps))  -- FastDiff_model.py:74-102.
 *   x     [B,1,T*256] device     mel [B,80,T] device     steps [B] device (float; fractional allowed, util.py:217)
 *   lens  [B] host, nullable: valid frames per utterance of a zero-padded batch (collate_2d, utils/__init__.py:136-150).
 *         NULL: the whole padded tensor is computed, as the reference does.  Given: utterance b is computed as if it were
 *         alone and lens[b] frames long -- the result in [0, lens[b]*256) is bit-identical to that single-utterance call,
 *         work behind it is skipped and the output there is unspecified.  (kernels = naive ignores lens.)
 *   eps_out [B,1,T*256] device, must not alias x.
 * Errors: T*256 length mismatch is the reference's assert at modules.py:236. */
FD_API int fd_forward(fd_handle h, const float *x, const float *mel, const float *steps, int B, int T,
                      const int *lens, float *eps_out, void *stream);

/* x_0 = sampling_given_noise_schedule(net, (B,1,T*256), dh, schedule, condition=mel, ddim, return_sequence)
 * -- util.py:158-235, called from FastDiff.py:101-103 and the notebooks.
 *   table   [N] host (see fd_step)
 *   x_T     [B,1,L] device, nullable: start noise.  NULL -> drawn on device (Philox4x32-10, `seed`).
 *   z       [N,B,1,L] device, nullable: z[k] is added after executed step k when table[k].add_noise
 *           (the reference draws std_normal on the CPU each step, util.py:63-68,229).  NULL -> Philox.
 *   out     [B,1,L] device result x_0.
 *   seq_out nullable, [N+1,B,1,L] device: x after each step, seq_out[0] = x_T (return_sequence=True, util.py:212-214,230-234).
 * The N-step loop is replayed from hipGraphs of up to 8 captured denoiser steps (kept per (B, T), 16 at most; step scalars are read
 * from a device table, so a graph does not depend on the schedule or the caller's pointers). */
FD_API int fd_sample(fd_handle h, const float *mel, int B, int T, const int *lens, const fd_step *table, int N,
                     int ddim, const float *x_T, const float *z, uint64_t seed, float *out, float *seq_out,
                     void *stream);

/* The range check of fd_sample (option "fallback" = "host", the default on both sides of the boundary): fd_sample enqueues only the
 * fp16x2 kernels; whether an operand left the fp16 range is then known on the HOST, after the work has run, and a C caller MUST call
 * fd_sample_check (or fd_sample_settle) before it reads `out`:
 *   fd_sample_check waits for the last fd_sample of this handle and, if one of its kernels raised a range flag, runs it again from
 *   the saved start with the flagged stages on their fp32 kernels.  Returns 1 if the call was redone, 0 if not, < 0 on error.
 * Until it has returned, `out` / `seq_out` of that fd_sample are provisional and its `z` must stay valid.  fd_forward, fd_sample (see
 * the pipelined form below), fd_commit_weights, fd_set_option, fd_read_tap and fd_destroy settle a pending check first, so nothing
 * is ever lost; fd_peak_normalize_int16[_ragged] and fd_mel_spectrogram do NOT (since round 3: they run on the provisional result
 * without waiting) -- a caller that reads `out`, or anything computed from it, must call fd_sample_check / fd_sample_settle before.  Schedules longer than 8 steps are checked (with a stream synchronisation) every 8 steps inside fd_sample.
 * With option "fallback" = "graph" (fp32 twins inside the graph) fd_sample_check is a no-op returning 0.
 *
 * Pipelined form (round 3; schedules of up to 8 steps, i.e. one graph launch per call): the next fd_sample on the handle does NOT
 * wait for the pending call -- it enqueues its own work first and looks at the previous call's flags afterwards, when the host's
 * wait falls on a busy GPU; a flagged call is then run again as a whole (its mel / x_T / z buffers must therefore stay valid until
 * it has been looked at, and its `out` is rewritten behind everything enqueued so far).  The waveform epilogue and the mel
 * front-end do not wait either.  A caller that enqueues work on a provisional `out` (epilogue, copies) asks afterwards:
 *   fd_sample_ticket(h)          the ticket of the last fd_sample on this handle (1, 2, ...);
 *   fd_sample_settle(h, ticket)  makes call `ticket` final (waits for it if nobody has looked at it yet) and returns 1 if it had to be
 *                                redone -- then whatever was computed from its `out` must be computed again -- 0 if not, < 0 on error.
 * Tickets older than the last 16 redone calls are reported as 0. */
FD_API int fd_sample_check(fd_handle h);
FD_API int64_t fd_sample_ticket(fd_handle h);
FD_API int fd_sample_settle(fd_handle h, int64_t ticket);

/* Per-utterance noise streams for the NEXT fd_sample call (one-shot; the reference draws std_normal per batch on the CPU,
 * util.py:63-68, so it has no counterpart there).  stream_ids [B] host: utterance b's x_T and z are then drawn from Philox stream
 * (seed, stream_ids[b]) with the counter running over the utterance's own samples -- the draw no longer depends on the position in
 * the batch or on the padded length, so an utterance gets the same waveform however a job is batched or sharded over GPUs
 * (fastdiff_amd/infer.py keys it on the utterance's index in the job).  Ignored for injected x_T / z. */
FD_API int fd_set_noise_streams(fd_handle h, const uint64_t *stream_ids, int B);

/* Waveform epilogue: wav/abs(wav).max() per utterance (FastDiff.py:110), *32767 -> int16
 * (utils/audio.py:11-16).  wav [B,1,L] device -> pcm [B,L] device int16. */
FD_API int fd_peak_normalize_int16(fd_handle h, const float *wav, int B, int64_t L, int16_t *pcm, void *stream);
/* The same for a zero-padded batch (fd_sample with lens): valid [B] host = samples of each utterance (lens[b]*256); the peak is
 * searched over the utterance's own samples only -- what FastDiff.py:110 sees for a batch of one -- and pcm behind them is 0.
 * valid NULL = the call above.  B <= 4096. */
FD_API int fd_peak_normalize_int16_ragged(fd_handle h, const float *wav, int B, int64_t L, const int64_t *valid, int16_t *pcm,
                                          void *stream);

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
 * build image, so its values are checked against an independent derivation only (tests/test_mel_frontend.py), not against librosa. */
FD_API int fd_set_mel_filterbank(fd_handle h, const float *fb, int n_mels, int n_bins);
FD_API int fd_get_mel_filterbank(fd_handle h, float *fb_out, int n_mels, int n_bins);

/* Options (key = value; the first value is the default).  Each one selects between code paths that ship tested; measured-and-rejected
 * variants are not options (LABBOOK.md keeps their numbers).
 *   "gemm" | "lvc" | "conv" = "f16x2" | "fp32"   the predictor GEMM / the LVC layers / DBlocks + ConvTranspose + predictor front on the fp16
 *                          matrix pipe with 2-piece operands (22 significant bits, fp32 accumulation) or on the exact-fp32 matrix instruction
 *   "lvc_h8"   = "mfma" | "valu"   hop-8 LVC layers on 16x16x32 fp16 tiles, or the all-VALU fp32 kernel (also their fp32 twin)
 *   "fallback" = "host" | "graph"  what happens when an operand does not fit fp16 in fd_sample.  host: only the fp16x2 kernels are enqueued,
 *                          their range flags are read on the host behind the work and a flagged call is run again on fp32 kernels
 *                          (fd_sample_check / fd_sample_settle make a result final: mandatory before `out` is read).  graph: every fp16x2
 *                          kernel is followed by its fp32 twin, which exits at once unless the flag is up (no host step, 21 more
 *                          launches per reverse step: +3 % at B = 8, +7 % at B = 1); fd_sample_check is then a no-op
 *   "hoist"    = "auto" | "on" | "off"   predict the kernels of all N <= 8 steps (or of each 8-step piece) with one front + GEMM launch pair
 *                          in front of the loop; auto: B * T <= 4096 frames
 *   "fuse_up" | "fuse_final" | "fuse_advance" | "embed_cache" = "1" | "0"   the block's ConvTranspose inside its first LVC layer (blocks 1, 2;
 *                          needs fallback = host) / final_conv inside the last LVC layer / the end-of-step bookkeeping inside the next
 *                          step's first kernel / the step-embedding rows kept between calls with the same schedule.  Same bits either way
 *   "mel"      = "pwg" | "tacotron"      which of the reference's two mel front-ends fd_mel_spectrogram computes
 *   "graph"    = "1" | "0"               replay the reverse loop from captured hipGraphs, or launch kernel by kernel
 * Test / measurement hooks: "kernels" = "fast" | "naive" and "kernels.<stage>" (embed, first, dblock, kp_front, kp_gemm, convt, lvc, final:
 * the one-thread-per-output kernel set), "taps" = "0" | "1" (keep block outputs for fd_read_tap), "profile" = "0" | "1" | "events"
 * (per-kernel timing, graph off), "lvc_dx" = "gather" | "copy" (training operator). */
FD_API int fd_set_option(fd_handle h, const char *key, const char *value);

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

/* Bookkeeping of the host-checked range fallback (option "fallback" = "host").  Names:
 *   "pieces"         8-step pieces the last fd_sample of more than 8 steps was enqueued as (0 for a shorter call);
 *   "pieces_redone"  of those, the pieces that raised a range flag and were run again from the saved x (settles a pending last piece);
 *   "pieces_fp32"    of those, the pieces enqueued with stages already on their fp32 kernels (after an earlier piece had flagged them);
 *   "fp32_mask"      the flag words (bit i = word i of "range_flags") those later pieces ran on fp32 -- sticky over the call;
 *   "calls_redone"   fd_sample calls of up to 8 steps run again as a whole since fd_create.
 * Returns the value (>= 0) or a negative status. */
FD_API int64_t fd_get_counter(fd_handle h, const char *name);

FD_API const char *fd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FASTDIFF_HIP_H */
