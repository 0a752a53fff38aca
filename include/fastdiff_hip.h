/*
 * fastdiff_hip.h -- C ABI of libfastdiff_hip.so: the MI355X (gfx950) FastDiff vocoder inference path.
 *
 * This is the drop-in boundary under the reference's Python API.  Every entry point names the reference interface it replaces
 * (path:line in Rongjiehuang/FastDiff).  Plain pointers and sizes only; no torch types.  All tensors are float32, contiguous,
 * [batch][channel][time] with time innermost -- the reference's own layout.
 * Companion headers: fastdiff_hip_ext.h (the rows next to the path -- int16 waveform epilogue, mel front-end -- and the test /
 * introspection hooks), fastdiff_hip_train.h (the training-side operators).
 *
 * Conventions
 *   - every function returns 0 on success, a negative fd_status on failure; fd_last_error() gives text.
 *   - a handle is bound to one device and is NOT thread safe (one handle per GPU, like one reference process per GPU under
 *     mp.spawn, utils/trainer.py:94-107).  Its calls share one workspace and are ordered by the stream they run on: a call on
 *     another stream than the previous one first waits (on the device) for the tail of that call, so consecutive calls may change
 *     streams but never overlap.  A stream may be destroyed once the calls made on it have returned (with option defer_check = 1:
 *     once they are settled).
 *   - device pointers are caller-owned; work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream).
 *     fd_forward never waits for the device.  fd_sample waits for its own work before it returns (see there) unless the caller
 *     opts out.  Other synchronisation: fd_create / fd_destroy / fd_commit_weights / fd_read_tap / workspace growth.
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

/* "fastdiff_hip <major.minor> (gfx950)"; fd_abi_revision: 2 = fd_sample settles its own range check before it returns unless option
 * defer_check = 1 (revision 1, library 0.1: fd_sample always returned a provisional result). */
FD_API const char *fd_version(void);
FD_API int fd_abi_revision(void);

/* Fills *cfg with the reference defaults above. */
FD_API int fd_default_config(fd_config *cfg);

/* FastDiff(**cfg).cuda(device): creates kernels' context on HIP device `device`.
 * Replaces: FastDiff.__init__ + .cuda() (FastDiff_model.py:13-72; FastDiff.py:17-29,  egs/demo.ipynb cell 0).
 * Any configuration the reference constructor accepts is taken.  base.yaml's architecture (the defaults above: every shipped YAML)
 * runs on the tuned gfx950 kernel set; any other one on runtime-shaped exact-fp32 kernels (a correctness path -- no graph, no
 * fd_read_tap).  FD_ERR_UNSUPPORTED only for what the reference's own forward / sampler cannot run: audio_channels != 1
 * (FastDiff_model.py:34), even lvc_kernel_size / kpnet_conv_size (modules.py:183-187,293-318), odd diffusion_step_embed_dim_in
 * (util.py:423), an upsample ratio < 2. */
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
 *   table   [N] host (see fd_step), 1 <= N <= 1024
 *   lens    as in fd_forward
 *   x_T     [B,1,L] device, nullable: start noise.  NULL -> drawn on device (Philox4x32-10, `seed`).
 *   z       [N,B,1,L] device, nullable: z[k] is added after executed step k when table[k].add_noise
 *           (the reference draws std_normal on the CPU each step, util.py:63-68,229).  NULL -> Philox.
 *   out     [B,1,L] device result x_0.
 *   seq_out nullable, [N+1,B,1,L] device: x after each step, seq_out[0] = x_T (return_sequence=True, util.py:212-214,230-234).
 * The loop is replayed from hipGraphs of up to 8 captured denoiser steps.  Step scalars, the caller's pointers and the utterance
 * lengths are read from device memory, and the library's own buffers are sized for T rounded up to a multiple of 32 frames
 * (option t_bucket) with the true lengths passed as `lens` -- so one graph serves every call of the same B whose T falls in the same
 * bucket, whatever the schedule (the reference CLI vocodes one utterance of a new length per call, FastDiff.py:97-103); the result is
 * bit-identical to the exact-T call.  Up to 64 graphs are kept (option graph_cache), least recently used evicted without a wait.
 *
 * By default fd_sample is "call, then read", like the reference: the contractions run on fp16 matrix instructions with 2-piece
 * operands (22 bits), every such kernel flags an operand outside the fp16 range, and before fd_sample returns it waits for its own
 * work, looks at those flags and -- rarely; never seen with weights in a trained model's range -- runs the call again with the
 * flagged stages on their exact-fp32 kernels.  `out` / `seq_out` are final (in stream order) when it returns.
 * Option fallback = "graph" keeps fd_sample fully asynchronous instead (every fp16x2 kernel is trailed by its fp32 twin inside the
 * graph; +3 ... +7 % time).  Option defer_check = "1" is the pipelined form below. */
FD_API int fd_sample(fd_handle h, const float *mel, int B, int T, const int *lens, const fd_step *table, int N,
                     int ddim, const float *x_T, const float *z, uint64_t seed, float *out, float *seq_out,
                     void *stream);

/* Pipelined range check -- OPT-IN: fd_set_option(h, "defer_check", "1") (fastdiff_amd's Python module and bench.py do).  fd_sample
 * then returns without waiting: `out` / `seq_out` are PROVISIONAL, and mel / x_T / z must stay valid, until the call is settled:
 *   fd_sample_check(h)           settles the last fd_sample: waits for it, redoes it on fp32 kernels if it raised a flag.
 *                                Returns 1 if it was redone, 0 if not, < 0 on error.
 *   fd_sample_ticket(h)          the ticket of the last fd_sample on this handle (1, 2, ...);
 *   fd_sample_settle(h, ticket)  makes call `ticket` final and returns 1 if it had to be redone -- then whatever the caller computed
 *                                from its `out` (epilogue, copies) must be computed again -- 0 if not, < 0 on error (tickets older than
 *                                the last 16 redone calls report 0).
 * The next fd_sample (schedules of up to 8 steps) enqueues its own work first and looks at the previous call's flags afterwards, so
 * the host's wait falls on a busy GPU.  fd_forward, fd_commit_weights, fd_set_option, fd_read_tap and fd_destroy settle a pending
 * call first; fd_peak_normalize_int16[_ragged] and fd_mel_spectrogram do not.  Schedules longer than 8 steps are checked every 8
 * steps inside fd_sample in either mode.  Without the option, and with fallback = "graph", nothing is ever pending: fd_sample_check and
 * fd_sample_settle return 0. */
FD_API int fd_sample_check(fd_handle h);
FD_API int64_t fd_sample_ticket(fd_handle h);
FD_API int fd_sample_settle(fd_handle h, int64_t ticket);

/* Per-utterance noise streams for the NEXT fd_sample call (one-shot; the reference draws std_normal per batch on the CPU,
 * util.py:63-68, so it has no counterpart there).  stream_ids [B] host: utterance b's x_T and z are then drawn from Philox stream
 * (seed, stream_ids[b]) with the counter running over the utterance's own samples -- the draw no longer depends on the position in
 * the batch or on the padded length, so an utterance gets the same waveform however a job is batched or sharded over GPUs
 * (fastdiff_amd/infer.py keys it on the utterance's index in the job).  Ignored for injected x_T / z. */
FD_API int fd_set_noise_streams(fd_handle h, const uint64_t *stream_ids, int B);

/* Options (key = value; the first value is the default).  Each one selects between code paths that ship tested; measured-and-rejected
 * variants are not options (LABBOOK.md keeps their numbers).
 *   "gemm" | "lvc" | "conv" = "f16x2" | "fp32"   the predictor GEMM / the LVC layers / DBlocks + ConvTranspose + predictor front on the
 *                          fp16 matrix pipe with 2-piece operands (22 significant bits, fp32 accumulation) or on the exact-fp32 one
 *   "gemm_form" = "winograd" | "direct"   the fp16x2 predictor GEMM evaluates kernel_conv's three taps as Winograd F(2,3) over the frame
 *                          axis (2/3 of the matrix work, same 2-piece arithmetic per product) or tap by tap
 *   "fallback" = "host" | "graph"  what fd_sample does about an operand outside the fp16 range: see fd_sample
 *   "defer_check" = "0" | "1"      fallback = host: settle inside fd_sample | the pipelined form (fd_sample_check / _settle)
 *   "t_bucket" = "32" | frames     fd_sample's buffers and graphs are sized for T rounded up to a multiple of this (0 = exact T)
 *   "graph_cache" = "64" | n       captured graphs kept per handle
 *   "graph"    = "1" | "0"         replay the reverse loop from captured hipGraphs, or launch kernel by kernel
 *   "hoist"    = "auto" | "on" | "off"   predict the kernels of all N <= 8 steps (or of each 8-step piece) with one front + GEMM launch
 *                          pair in front of the loop; auto: B * T <= 4096 frames
 *   "lvc_h8"   = "mfma" | "valu"   hop-8 LVC layers on 16x16x32 fp16 tiles, or the all-VALU fp32 kernel (also their fp32 twin)
 *   "fuse_up" | "fuse_final" | "fuse_advance" | "embed_cache" = "1" | "0"   the block's ConvTranspose inside its first LVC layer
 *                          (blocks 1, 2; needs fallback = host) / final_conv inside the last LVC layer / the end-of-step bookkeeping
 *                          inside the next step's first kernel / the step-embedding rows kept between calls.  Same bits either way
 *   "mel"      = "pwg" | "tacotron"      which of the reference's two mel front-ends fd_mel_spectrogram computes (fastdiff_hip_ext.h)
 * Test / measurement hooks: "kernels" = "fast" | "naive" and "kernels.<stage>" (embed, first, dblock, kp_front, kp_gemm, convt, lvc,
 * final: the one-thread-per-output kernel set), "taps" = "0" | "1" (keep block outputs for fd_read_tap), "profile" = "0" | "1" |
 * "events" (per-kernel timing, graph off), "lvc_dx" = "gather" | "copy" (training operator). */
FD_API int fd_set_option(fd_handle h, const char *key, const char *value);

#ifdef __cplusplus
}
#endif
#endif /* FASTDIFF_HIP_H */
