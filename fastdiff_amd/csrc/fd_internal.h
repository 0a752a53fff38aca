// fd_internal.h -- shared between the host side (fd_api.cpp, fd_weights.cpp) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/fastdiff_hip.h"
#include "../../include/fastdiff_hip_ext.h"
#include "../../include/fastdiff_hip_train.h"

// ---------------------------------------------------------------------------------------------
// Fixed architecture of this build (modules/FastDiff/config/base.yaml:21-33).  fd_create rejects
// anything else: the four dataset YAMLs of the reference never change the network shape.
// ---------------------------------------------------------------------------------------------
namespace fd {
constexpr int C = 32;          // inner_channels
constexpr int COND = 80;       // cond_channels
constexpr int HID = 64;        // kpnet_hidden_channels
constexpr int NBLK = 3;
constexpr int LAYERS = 4;      // lvc_layers_each_block
constexpr int HOPT = 256;      // prod(upsample_ratios)
constexpr int E_IN = 128, E_MID = 512, E_OUT = 512;
constexpr int KLAYER = 2 * C * C * 3;        // 6144 predicted-kernel floats per (frame, layer)
constexpr int KW = KLAYER * LAYERS;          // 24576
constexpr int KB = 2 * C * LAYERS;           // 256 predicted biases per frame
constexpr int KREC = KW + KB;                // 24832 floats: one frame's packed record
__host__ __device__ constexpr int ratio(int n) { return n == 2 ? 4 : 8; }        // upsample_ratios [8,8,4]
__host__ __device__ constexpr int hop(int n) { return n == 0 ? 8 : (n == 1 ? 64 : 256); }
__host__ __device__ constexpr int down_factor(int d) { return d == 0 ? 4 : 8; }   // reversed ratios (FastDiff_model.py:63)

// Packed position of predicted-kernel element K[layer][in][out][tap] inside a frame record.
// The record of one (frame, layer) is laid out as the A operand of v_mfma_f32_32x32x2_f32:
//   [mt][s4 = step/4][lane][r = step%4],  step = kk/2, lane = row + 32*(kk%2), kk = tap*32 + in
// so a lane fetches four consecutive k-steps of its row with one 16-byte load.  The 64 output channels are
// split over the two 32-row tiles so that each tile is closed under the gate (modules.py:217): gate channel
// ch = out%32 has its sigmoid input (out < 32) and its tanh input (out >= 32) in the SAME tile, 16 rows apart:
//   mt = ch/16,  row = ch%16 + 16*(out/32).
// In the MFMA result a lane then holds both halves of a channel (registers r and r+8), so sigmoid*tanh needs no
// cross-lane traffic, and for hop 256 a wave needs only ONE of the two tiles (48 instead of 96 operand registers).
__host__ __device__ inline int kernel_row(int out) { return (out & 15) + 16 * (out >> 5); }
__host__ __device__ inline int kernel_tile(int out) { return (out & 31) >> 4; }
__host__ __device__ inline int kernel_index(int layer, int in, int out, int tap)
{
    // A operand of a 32x32x16 MFMA: lane = row + 32*g holds the 8 consecutive k = 16*kg + 8*g + e, k = tap*32 + in
    const int kk = tap * C + in, kg = kk >> 4, g = (kk >> 3) & 1, e = kk & 7;
    const int mt = kernel_tile(out), lane = kernel_row(out) + 32 * g;
    return layer * KLAYER + ((mt * 6 + kg) * 64 + lane) * 8 + e;
}
// Position of predicted bias (layer, out): same [mt][row] order as the kernel rows.
__host__ __device__ inline int bias_index(int layer, int out) { return KW + layer * 64 + kernel_tile(out) * 32 + kernel_row(out); }
}  // namespace fd

// ---------------------------------------------------------------------------------------------
// Device-side weights
// ---------------------------------------------------------------------------------------------
struct ConvW {
    const float *w = nullptr;   // folded, reference layout [out][in][k]  (ConvTranspose: [in][out][k]; Linear: [out][in])
    const float *b = nullptr;
};

struct DevWeights {
    // reference-layout (naive kernels, VALU kernels)
    ConvW first, final_;
    struct { ConvW res, conv[3]; } down[fd::NBLK];
    struct { ConvW fc_t, up, kp_in, kp_res[6], kc, bc, convs[fd::LAYERS]; } blk[fd::NBLK];
    // embed MLP, transposed [in][out] so thread-per-output reads are coalesced
    const float *fc_t1_T = nullptr, *fc_t1_b = nullptr, *fc_t2_T = nullptr, *fc_t2_b = nullptr;
    const float *fc_t_T[fd::NBLK] = {}, *fc_t_b[fd::NBLK] = {};
    const float *embed_table = nullptr;           // [64] fp32 frequencies
    // MFMA A-operand packs ([mt][s4][lane][4], kk = tap*Cin + in)
    const float *down_pack[fd::NBLK][4] = {};     // conv0..2 (K=96), res 1x1 (K=32)
    const uint16_t *down_h2[fd::NBLK][4] = {};    // the same as two fp16 pieces: [piece][kg][64 lane][8]
    bool dblock_f16_ok = false;
    const float *lvc_conv_pack[fd::NBLK][fd::LAYERS] = {};
    const uint16_t *lvc_conv_h2[fd::NBLK][fd::LAYERS] = {};   // the same as two fp16 pieces: [piece][6 kg][64 lane][8], k = tap*32 + in
    const uint16_t *lvc_conv_h16[fd::LAYERS] = {};        // block 0 (hop 8): A operands of v_mfma_f32_16x16x32_f16: [row tile 2][tap 3][piece 2][64 lane][8]
    bool lvc_f16_ok = false;                      // every LVC conv weight fits the fp16 range
    const float *kp_in_pack[fd::NBLK] = {};       // 80->64 k5: 2 mt x 50 s4
    const float *kp_res_pack[fd::NBLK][6] = {};   // 64->64 k3: 2 mt x 24 s4
    const uint16_t *kp_in_h2[fd::NBLK] = {};      // the same as fp16 pieces: [2 mt][piece][25 kg][64 lane][8]
    const uint16_t *kp_res_h2[fd::NBLK][6] = {};  //                          [2 mt][piece][12 kg][64 lane][8]
    bool kpf_f16_ok = false;
    const float *gemm_pack[fd::NBLK] = {};        // kernel_conv+bias_conv as MFMA B operand: [776 ptile][24 s4][64][4]
    const float *gemm_bias[fd::NBLK] = {};        // [24832] conv biases in packed order
    const uint16_t *gemm_h2_pack[fd::NBLK] = {};  // same weights as two fp16 pieces (w1, (w-w1)*2^11): [776 ptile][2][12 kg][64 lane][8]
    bool gemm_f16_ok = false;                     // every GEMM weight fits the fp16 range
    // the same weights transformed for Winograd F(2,3) over the frame axis of kernel_conv (k_kp_gemm_w): per packed column four K = 64
    // blocks g0, (g0+g1+g2)/2, (g0-g1+g2)/2, -g2 (formed in double, rounded to fp32, split): [776 ptile][2 pieces][16 kg][64 lane][8]
    const uint16_t *gemm_w_pack[fd::NBLK] = {};
    bool gemm_w_ok = false;                       // ... and every transformed weight fits the fp16 range
    const float *up_pack[fd::NBLK] = {};          // ConvTranspose as per-phase A operands: [r phases][8 s4][64][4]
    const uint16_t *up_h2[fd::NBLK] = {};         // ConvTranspose per-phase slices as fp16 pieces: [ph][piece][4 kg][64 lane][8]
    bool convt_f16_ok = false;
    const float *final_fuse = nullptr;            // final_conv weights in the last LVC layer's register order: [mt*2 + hi][8 r][8] (7 taps + pad)
    const int *kc_perm = nullptr;                 // [24576] reference kernel_conv row -> packed position
    const int *bc_perm = nullptr;                 // [256] reference bias_conv row -> position inside the bias part
};

// ---------------------------------------------------------------------------------------------
// Per-call step description read by kernels from device memory (so a captured graph is pointer-free)
// ---------------------------------------------------------------------------------------------
struct StepParams {
    fd_step table[1024];
    const float *z;        // injected noise [N,B,L] or null
    float *seq;            // trajectory [N+1,B,L] or null
    unsigned long long seed;
    int n_steps;
    int ddim;
    int step_idx;          // advanced on device at the end of every captured step
    int l4;                // samples / 4 of one (padded) utterance of this call: splits a flat float4 index into (utterance, offset)
    const unsigned long long *uids;   // [B] per-utterance noise stream ids (fd_set_noise_streams) or null: see philox_normal4
    int l4_io;             // samples / 4 of one utterance as the CALLER lays it out (x_T, z, seq_out, out): == l4 unless the call's frames were
                           // rounded up to a bucket (fd_context::t_bucket); also what a Philox draw's position is counted in
    long long n4_io;       // B * l4_io: one step's stride in z / seq_out
};

enum Stage { ST_EMBED = 0, ST_FIRST, ST_DBLOCK, ST_KP_FRONT, ST_KP_GEMM, ST_CONVT, ST_LVC, ST_FINAL, ST_COUNT };

struct Workspace {
    int B = 0;                  // capacity: utterances (per-utterance arrays),
    int64_t frames = 0;         //           B*T frames (activations, predicted kernels),
    int64_t rows = 0;           //           B*gx_rows(T) rows of the GEMM's fp16 image
    float *noise = nullptr;     // [1024][B][3][80]
    float *embed_h2 = nullptr;  // [max(1024, B)][512] second MLP layer of the step embedding, between the two embed kernels
    float *a[4] = {};           // a0..a3
    float *kp_h0 = nullptr, *kp_hA = nullptr, *kp_hB = nullptr;   // [3][B][64][T]
    float *kpack = nullptr;     // [3][B][T][KREC]
    float *h_f16 = nullptr;     // fp16 piece image of the predictor hidden state: [3][B][64*ceil(T/64)+2 rows][2 pieces][64] x 2 B
    int *lens_dev = nullptr;    // [B] valid frames per utterance of the current call (ragged batches)
    unsigned long long *uid_dev = nullptr;   // [B] noise stream ids of the current call (fd_set_noise_streams)
    float *xsave = nullptr;     // [B][L] x at the start of the call / of the current graph chunk (option fallback = host: what a redo starts from)
    int *range_flag = nullptr;  // (128 words) [0] predictor GEMM, [1 + 4*block + layer] LVC layers, [13 + d] DBlocks,
                                // [16 + n] ConvTranspose of block n, [19] predictor front: an operand did not fit fp16; 32 words, zeroed every step;
                                // words 32..63: the same flags of the previous sampler step (skip_after_previous_overflow);
                                // words 64..95: OR over all steps since the last clear (what option fallback = host reads back)
    float *xA = nullptr, *xB = nullptr;                           // [B][32][L] ping-pong
    float *xtap[fd::NBLK] = {}; // block outputs kept for fd_read_tap
    int64_t pframes = 0, prows = 0, plens = 0;   // capacity of the predictor's buffers (kp_h*, kpack, mel_rep / h_f16 / lens_dev): frames, image
                                                 // rows and utterances x the reverse steps predicted at once (fd_context::hoist_np), tracked on
                                                 // their own so that a large batch and a hoisted small one do not multiply
    float *mel_rep = nullptr;   // [N][B][80][T] the mel once per reverse step: the hoisted predictor's batch
    float *mel = nullptr;       // [B][80][T] library-owned copy used by the sampler graph
    float *x = nullptr;         // [B][L] running x_t of the sampler
    float *eps_acc = nullptr;   // [B][L] final_conv sums written by the last LVC layer (k_lvc_h2 FINAL); all zero between steps
    float *steps = nullptr;     // [B] step values for fd_forward
    StepParams *params = nullptr;
    size_t bytes = 0;
};

struct ProfEntry { std::string name; hipEvent_t e0, e1; };

// Constant tables of the mel front-end (fd_kernels_mel.hip), built in double precision at fd_create.
struct MelTables {
    const float *tab = nullptr;      // cos[1024], sin[1024] of 2*pi*i/1024, periodic Hann window [1024]
    const int *fb_lo = nullptr, *fb_n = nullptr, *fb_off = nullptr;      // per mel filter: first FFT bin, #bins, offset into fb_w
    const float *fb_w = nullptr;     // the non-zero weights of librosa.filters.mel(22050, 1024, 80, fmin, fmax), filter after filter
};
enum MelVariant { MEL_PWG = 0, MEL_TACOTRON = 1, MEL_VARIANTS = 2 };

// A parameter as fd_commit_weights folds it (weight-norm applied, reference layout) and the configuration-generic path that takes them
// (fd_generic.hip: any architecture the reference constructor accepts, on runtime-shaped kernels; gen == nullptr for base.yaml's
// architecture, which runs on the tuned kernel set).
struct FoldedParam { std::vector<float> w, b; };
namespace fdg {
struct Net;
int validate(const fd_config &c, std::string &why);
int create(fd_context *c);
void destroy(fd_context *c);
int hop_total(const fd_context *c);
int commit(fd_context *c, const std::map<std::string, FoldedParam> &f);
int forward(fd_context *c, const float *x, const float *mel, const float *steps, int B, int T, const int *lens, float *eps_out, hipStream_t stream);
int sample(fd_context *c, const float *mel, int B, int T, const int *lens, const fd_step *table, int N, int ddim, const float *x_T, const float *z,
           unsigned long long seed, const std::vector<unsigned long long> &ids, float *out, float *seq_out, hipStream_t stream);
}  // namespace fdg

struct fd_context {
    fd_config cfg;
    fdg::Net *gen = nullptr;                  // set for a configuration other than base.yaml's: every compute call goes to fd_generic.hip
    int device = 0;
    int num_cus = 256;
    std::string err;
    bool committed = false;
    bool fast[ST_COUNT];
    bool use_graph = true;
    int profile = 0;                          // option "profile": 0 off | 1 the kernels' own begin / end timestamps | 2 ("events") stream events around each launch
    bool keep_taps = false;
    bool gemm_f16 = true;                     // kp_gemm on the fp16 matrix pipe with the 2-piece operand split
    bool gemm_wino = true;                    // option "gemm_form" = "winograd" | "direct": ... as Winograd F(2,3) over the frame axis (2/3 of the matrix work)
    bool lvc_f16 = true;                      // LVC layers (hop 64, 256) likewise
    bool conv_f16 = true;                     // DBlocks, ConvTranspose upsamplers and the predictor front likewise
    const int *step_lens = nullptr;           // device copy of the caller's `lens` for this call (ragged batch), or null
    std::vector<unsigned long long> noise_ids; // fd_set_noise_streams: consumed by the next fd_sample
    // How a stage that has an fp16x2 kernel is launched (fd_pipe below):
    //   inline_fallback  the fp32 kernel is enqueued right behind the fp16x2 one and exits at once unless that one raised its flag
    //                    (fully asynchronous, works inside a captured graph; ~2 us per stage and step)
    //   !inline_fallback only the fp16x2 kernel; flags accumulate in range_flag[64..95] and the HOST redoes the work with
    //                    fp32_mask set (option fallback = host: fd_sample_check)
    //   fp32_mask        bit i set: the stage whose flag word is i runs its fp32 kernel outright
    bool lvc_h8_mfma = true;                  // option "lvc_h8" = "mfma": the hop-8 layers on 16x16x32 fp16 tiles (k_lvc_h8m) | "valu" (k_lvc_h8)
    bool host_fallback = true;                // option "fallback" = "host" (default; settled inside fd_sample unless defer_check) | "graph"
    bool inline_fallback = true;
    unsigned fp32_mask = 0;
    bool h_image_ready = false;               // set by fast_kp_front when it wrote the GEMM's fp16 image of h for this step
    std::map<std::string, std::pair<std::vector<int64_t>, std::vector<float>>> raw;   // host copies from fd_set_weight
    std::vector<void *> dev_allocs;          // weight arena pieces
    DevWeights w;
    Workspace ws;
    bool fuse_final = true;                  // option "fuse_final": final_conv inside the last LVC layer
    bool fuse_up = true;                     // option "fuse_up": the ConvTranspose of blocks 1 and 2 inside their first LVC layer (when both
                                             // stages run fp16x2-only, i.e. under fallback = host or a forced mask without them).  Same
                                             // bits, one launch and one round trip of x less per block: B=1 -5.2 %, B=8 -2.6 %
    bool final_fused = false;                // set by the last LVC layer's launch, consumed by fast_final
    bool fuse_advance = true;                // option "fuse_advance": between two steps of one graph / launch sequence the end-of-step
                                             // bookkeeping (k_advance) rides in the next step's first kernel instead of a launch of its own
    bool lvc_dx_gather = true;               // option lvc_dx = gather | copy: the frames path's dx kernel reads kernel_conv's frames (fd_kernels_train.hip)
    bool advance_pending = false;            // set by enqueue_steps after a step whose bookkeeping the next first_conv will do
    // the step embedding and the three fc_t rows of every reverse step depend on the schedule's t values and the weights only: kept from
    // the previous fd_sample when those are unchanged (two launches per call)
    std::vector<float> embed_t;
    int embed_B = 0;
    bool embed_valid = false;
    bool embed_cache = true;                 // option "embed_cache"
    MelTables mel[MEL_VARIANTS];             // [MEL_PWG]: fmin 80, fmax 7600; [MEL_TACOTRON]: fmin 0, fmax 8000 (twiddles/window shared)
    int mel_variant = MEL_PWG;               // option "mel"
    std::vector<float> mel_bank[MEL_VARIANTS];   // the dense [80][513] bank each front-end uses (host copy: fd_get_mel_filterbank)
    bool mel_bank_user[MEL_VARIANTS] = {false, false};   // supplied through fd_set_mel_filterbank (else the restated default)
    std::vector<void *> mel_allocs;           // device memory behind `mel` (built on first use, freed at fd_destroy)
    int last_B = 0, last_T = 0;
    hipStream_t cap_stream = nullptr;
    // Calls on one handle share the workspace, the embedding rows and the pending range check: they are ordered by the stream they run
    // on.  When a caller moves to another stream (fd_forward / fd_sample), a pending check is settled on the old stream and the new
    // stream waits for everything the old one still holds (follow_stream in fd_api.cpp).
    hipStream_t last_stream = nullptr;
    bool have_last_stream = false;
    hipEvent_t ev_switch = nullptr;          // the tail of the handle's last compute call (fd_api.cpp: mark_tail / follow_stream)
    bool tail_marked = false;
    // The predictor (front + GEMM) sees the mel and the step embedding only -- never x -- so for a short schedule on a small batch all N
    // steps' kernels are predicted by ONE pair of launches in front of the loop (batch entry n * B + b = step n of utterance b): at
    // B = 1 the front's seven-layer latency chain and the GEMM's fill are paid once per call instead of once per step.  hoist_np = N
    // while such a call is enqueued (1 otherwise), hoist_step = the step being enqueued: the LVC layers read kpack at that offset.
    // A longer schedule does the same per captured piece of 8 steps (hoist_chunk): the piece's graph starts with the predictor of its
    // own steps (rows step_idx .. step_idx + 7 of the embedding table, read through the device step counter).
    // option "hoist" = auto (B * T <= 4096 frames, N >= 2) | on | off
    int hoist_mode = 1;                       // 0 off, 1 auto, 2 on
    int hoist_np = 1, hoist_step = 0;
    bool hoist_chunk = false;                 // a long schedule (N > 8): the predictor of each 8-step graph piece is hoisted to the piece's front
    // captured denoiser steps, one per (B, T, mode): micro-batches of different padded length alternate without re-capturing
    struct StepGraph { int B, T, steps; unsigned sig; hipGraph_t graph; hipGraphExec_t exec; unsigned long long last_use; };   // `steps` denoiser steps per launch
    std::vector<StepGraph> graphs;           // at most max_graphs, least recently used evicted
    unsigned long long graph_clock = 0;
    // An evicted graph may still be running: it is parked here with an event recorded behind everything enqueued so far and destroyed
    // by a later call once that event has completed (no stream-wide wait on the eviction path).
    struct RetiredGraph { hipGraph_t graph; hipGraphExec_t exec; hipEvent_t done; };
    std::vector<RetiredGraph> retired;
    int max_graphs = 64;                     // option "graph_cache"
    // The reference CLI vocodes one utterance per call with a different length each time (FastDiff.py:97-103, base.yaml:53): a graph
    // keyed on the exact T would be captured once per utterance.  fd_sample therefore rounds the frames of its OWN buffers up to a
    // multiple of t_bucket (option "t_bucket", 0 / 1 = off) and runs the call with `lens` -- whose contract already is "as if the
    // utterance were alone and lens[b] frames long", bit for bit -- so one graph serves every length of a bucket; the caller's
    // tensors keep their own dense layout (StepParams::l4_io).
    int t_bucket = 32;
    long long n_graph_captures = 0, n_graph_hits = 0, n_graph_evictions = 0;
    bool defer_check = false;                // option "defer_check": fd_sample returns with its range check still pending (tickets)
    // Pinned staging ring for everything a call uploads from the host (step table, lens, noise stream ids, valid counts): a slot
    // is rewritten only after the event recorded behind its last upload has completed, so no call waits for the stream and no
    // asynchronous copy ever reads memory the next call has already overwritten.
    static constexpr int STAGE_SLOTS = 8;
    struct StageSlot { char *host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; };
    StageSlot stage[STAGE_SLOTS];
    unsigned stage_next = 0;
    // Everything fd_sample was called with (host arrays copied): what a redo of a call of up to 8 steps starts from again
    struct SampleArgs {
        const float *mel = nullptr;
        int B = 0, T = 0, N = 0, ddim = 0;
        bool has_lens = false;
        std::vector<int> lens;
        std::vector<fd_step> table;
        const float *x_T = nullptr, *z = nullptr;
        unsigned long long seed = 0;
        float *out = nullptr, *seq_out = nullptr;
        hipStream_t stream = nullptr;
        std::vector<unsigned long long> ids;
    };
    // option fallback = host: the fd_sample call whose range flags have not been looked at yet (fd_sample_check / fd_sample_settle).
    //   lazy (schedules of up to 8 steps = one graph launch): the NEXT fd_sample enqueues its own work first and looks at this call's
    //   flags afterwards, so the host's wait falls on a busy GPU; a flagged call is then run again as a whole from `args`;
    //   otherwise (longer schedules, checked every 8 steps): redone from `xsave`, steps [first, first + count).
    struct PendingCall {
        bool active = false;
        bool lazy = false;
        int slot = 0;                                       // which of the two flag buffers / events
        long long ticket = 0;
        int B = 0, T = 0, N = 0, first = 0, count = 0;      // steps [first, first + count) were enqueued without fallbacks; T: frames of the
        int T_io = 0;                                       // library's buffers (bucketed), T_io: the caller's
        float *out = nullptr;
        hipStream_t stream = nullptr;
        SampleArgs args;
    } pending;
    long long ticket_counter = 0;            // one per fd_sample
    // fd_get_counter: the last long call's 8-step pieces (enqueued / run again after a flag / enqueued with stages already on fp32 and
    // with which), and the short calls run again as a whole since fd_create
    long long n_pieces = 0, n_pieces_redone = 0, n_pieces_fp32 = 0, n_calls_redone = 0;
    unsigned call_fp32_mask = 0;
    long long redone_ring[16] = {};          // tickets of the calls that had to be redone (0 = none), newest overwrite oldest
    unsigned redone_next = 0;
    int *flags_host = nullptr;               // pinned, 2 x 32 words: the sticky flags of the pending call(s)
    hipEvent_t flags_done = nullptr, flags_done2 = nullptr;
    void *scratch = nullptr;                 // 64 KB device scratch (abs-max words, ...)
    float *lvc_scratch = nullptr;            // the LVC operator's frame-major kernel copy (fd_lvc_forward / fd_lvc_backward), grown on demand
    size_t lvc_scratch_bytes = 0;
    float *kconv_scratch = nullptr;          // the row-slice partial sums of fd_kconv_backward's dh pass, grown on demand
    size_t kconv_scratch_bytes = 0;
    float *cconv_scratch = nullptr;          // per-workgroup partial sums of fd_conv32_backward's dW / db, grown on demand
    size_t cconv_scratch_bytes = 0;
    std::vector<ProfEntry> prof_pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, std::pair<int64_t, double>> prof_acc;
};

enum Pipe { PIPE_F16_THEN_F32, PIPE_F16_ONLY, PIPE_F32_ONLY };
inline Pipe fd_pipe(const fd_context *c, bool f16_possible, int flag_word)
{
    if (!f16_possible || ((c->fp32_mask >> flag_word) & 1u)) return PIPE_F32_ONLY;
    return c->inline_fallback ? PIPE_F16_THEN_F32 : PIPE_F16_ONLY;
}

// Arguments of one denoiser step (all device pointers)
struct StepIO {
    const float *x_in;     // [B][L] current x_t
    const float *mel;      // [B][80][T]
    const float *steps;    // [B] (forward mode) -- ignored in sampler mode (table[step_idx].t)
    float *eps_out;        // forward mode: eps destination; sampler mode: null (x updated in place)
    int sampler;           // 0 = fd_forward, 1 = fd_sample step
};

// kernels (fd_kernels_*.hip) -- each returns hipError from launch
namespace fdk {
struct Launch {
    fd_context *ctx;
    hipStream_t stream;
    bool capturing;
};
hipError_t embed(const Launch &L, const StepIO &io, int B, int n_steps);
hipError_t first_conv(const Launch &L, const StepIO &io, int B, int T);
hipError_t dblock(const Launch &L, const StepIO &io, int d, int B, int T);
hipError_t kp_front(const Launch &L, const StepIO &io, int B, int T);
hipError_t kp_gemm(const Launch &L, int B, int T);
hipError_t advance_step(const Launch &L);
hipError_t clear_range_flags(const Launch &L, int set_step = 0);     // before the first step of a call / of a redo: step counter := set_step
hipError_t mel_frontend(const Launch &L, const float *wav, int B, int64_t n_samples, float *mel, int T);
hipError_t init_noise(const Launch &L, float *x, int B, int l4, int l4_io, unsigned long long seed, const unsigned long long *uids);
hipError_t copy_rows(const Launch &L, float *dst, int64_t dpitch, const float *src, int64_t spitch, int width, int rows, int reps = 1,
                     int64_t rep_stride = 0);
hipError_t peak_normalize_int16(const Launch &L, const float *wav, int B, int64_t len, int16_t *pcm, const long long *valid_dev);
}  // namespace fdk

// profiling-aware launch helper.  Option profile = 1: the launch goes through hipExtLaunchKernelGGL with a start and a stop event, which
// receive the dispatch's own begin / end timestamps (what rocprofv3 --kernel-trace reports) -- nothing is put between two launches, so
// the kernels run back to back as in the replayed graph.  profile = events: hipEventRecord in front of and behind each launch (two
// barrier packets per kernel: the gaps let the previous kernel's dirty lines drain, the numbers come out 3-5 % shorter).
bool fd_prof_stamps(const fdk::Launch &L, const char *name, hipEvent_t *e0, hipEvent_t *e1);
void fd_prof_begin(const fdk::Launch &L, const char *name);
void fd_prof_end(const fdk::Launch &L);

#define FD_LAUNCH(L, name, kernel, grid, block, shmem, ...)                                  \
    do {                                                                                     \
        hipEvent_t ev0__ = nullptr, ev1__ = nullptr;                                         \
        if (fd_prof_stamps(L, name, &ev0__, &ev1__)) {                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, (L).stream, ev0__, ev1__, 0, __VA_ARGS__); \
        } else {                                                                             \
            fd_prof_begin(L, name);                                                          \
            hipLaunchKernelGGL(kernel, grid, block, shmem, (L).stream, __VA_ARGS__);         \
            fd_prof_end(L);                                                                  \
        }                                                                                    \
        hipError_t e__ = hipGetLastError();                                                  \
        if (e__ != hipSuccess) return e__;                                                   \
    } while (0)
