// fd_generic.hip -- the denoiser and the reverse loop for ANY configuration the reference constructor accepts
// (modules/FastDiff/module/FastDiff_model.py:13-26: inner_channels, cond_channels, upsample_ratios, lvc_layers_each_block,
// lvc_kernel_size, kpnet_hidden_channels, kpnet_conv_size, diffusion_step_embed_dim_in / _mid / _out), built by
// FastDiffTask.build_model from hparams (modules/FastDiff/task/FastDiff.py:17-29).
//
// The tuned kernel set (fd_kernels_*.hip) exists for base.yaml's architecture only -- the one every shipped YAML uses.  A handle
// created with another configuration runs here: runtime-shaped kernels, one thread per output element, fp32 multiply-adds in a fixed
// order, tensors in the reference's own layouts ([B, C, time]; the predicted kernels as kernel_conv leaves them, [B, layers*C*2C*ks, T]
// with T innermost: modules.py:333-338).  Correctness path, not a fast one: nothing here is tiled, staged through LDS or captured in a
// graph; fd_read_tap / profiling are not offered.  `lens` means what it means on the tuned path: utterance b is computed as if it were alone
// and lens[b] frames long (every kernel treats positions behind it as the zero padding at the end of a signal and skips the outputs there).
#include "fd_internal.h"
#include "fd_device.h"
#include "fd_kernels.h"

#include <cmath>

namespace fdg {

struct Conv { const float *w = nullptr, *b = nullptr; };

struct Net {
    fd_config cfg;
    int nb = 0, hop_total = 1;
    int hop[8] = {0};                  // cond_hop_length of block n (cumulative product of the ratios, FastDiff_model.py:47-49)
    Conv first, final_, fc_t1, fc_t2;
    struct Blk {
        Conv fc_t, up, kp_in, kp_res[6], kc, bc, res, dconv[3];
        std::vector<Conv> convs;
    } blk[8];
    const float *embed_table = nullptr;
    std::vector<void *> allocs;        // weights
    // workspace, grown to the largest call seen
    float *ws = nullptr;
    size_t ws_floats = 0;
};

// ---------------------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act(float v, float slope) { return v >= 0.0f ? v : v * slope; }      // slope 1 = identity

// calc_diffusion_step_embedding (util.py:407-432): emb[b] = cat(sin(t_b * e), cos(t_b * e)), e from the host table (fp32 product, fp32 exp)
__global__ void g_step_embed(const float *steps, float t_all, const float *table, int half, float *emb, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    const float arg = (steps ? steps[b] : t_all) * table[j];
    emb[(int64_t)b * 2 * half + j] = sinf(arg);
    emb[(int64_t)b * 2 * half + half + j] = cosf(arg);
}

// out[b][n] = f(bias[n] + sum_k W[n][k] * in[b][k]);  swish: f(v) = v * sigmoid(v)  (FastDiff_model.py:7-8,85-87; modules.py:202)
__global__ void g_linear(const float *in, const float *W, const float *bias, float *out, int B, int K, int N, int swish)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, n = i - b * N;
    float acc = bias[n];
    for (int k = 0; k < K; ++k) acc += W[(int64_t)n * K + k] * in[(int64_t)b * K + k];
    out[i] = swish ? acc / (1.0f + expf(-acc)) : acc;
}

// y[b][o][t] = post(bias[o] + sum_{i,k} W[o][i][k] * pre(xin(b, i, t + k*dil - pad))) (+ add[b][o][t*add_stride])
//   xin(b, i, p) = 0 outside [0, Lout), else x[b][i][p * in_stride] (+ in_add[b][i]: the predictor's `c + noise`, modules.py:203 --
//   added to the signal, not to its zero padding).  in_stride > 1 reads every in_stride-th sample: nearest-neighbour down-sampling by
//   an integer factor (DiffusionDBlock, modules.py:127-134) without materialising the picked sequence.
//   lens (nullable) / spf: utterance b is lens[b] * spf output samples long; behind that nothing is computed and nothing is read.
__global__ void g_conv1d(const float *x, const float *W, const float *bias, float *y, int B, int Cin, int Cout, int K, int dil, int pad,
                         int64_t Lx, int64_t Lout, int in_stride, const float *in_add, float pre, float post, const float *add,
                         int64_t Ladd, int add_stride, const int *lens, int spf)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * Cout * Lout) return;
    const int64_t t = i % Lout;
    const int o = (int)((i / Lout) % Cout), b = (int)(i / (Lout * Cout));
    const int64_t Lb = lens ? (int64_t)lens[b] * spf : Lout;
    if (t >= Lb) return;
    float acc = bias[o];
    for (int c = 0; c < Cin; ++c) {
        const float *xr = x + ((int64_t)b * Cin + c) * Lx;
        const float ia = in_add ? in_add[(int64_t)b * Cin + c] : 0.0f;
        for (int k = 0; k < K; ++k) {
            const int64_t p = t + (int64_t)k * dil - pad;
            if (p >= 0 && p < Lb) acc += W[((int64_t)o * Cin + c) * K + k] * act(xr[p * in_stride] + ia, pre);
        }
    }
    acc = act(acc, post);
    if (add) acc += add[((int64_t)b * Cout + o) * Ladd + t * add_stride];
    y[i] = acc;
}

// ConvTranspose1d(C, C, 2r, stride r, padding r/2 + r%2, output_padding r%2) of leaky_relu(x, 0.2) (modules.py:163-166,205-206):
// y[o][t] = b[o] + sum_i sum_j lrelu(x[i][j]) * W[i][o][t + p - j*r] over the j with 0 <= t + p - j*r < 2r; out length = r * in length
__global__ void g_convt(const float *x, const float *W, const float *bias, float *y, int B, int C, int r, int pad, int64_t Lin,
                        const int *lens, int spf_in)
{
    const int64_t Lo = Lin * r;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C * Lo) return;
    const int64_t t = i % Lo;
    const int o = (int)((i / Lo) % C), b = (int)(i / (Lo * C));
    const int64_t Lib = lens ? (int64_t)lens[b] * spf_in : Lin;      // this utterance's input length
    if (t >= Lib * r) return;
    float acc = bias[o];
    const int64_t q = t + pad, j_hi = q / r;      // the kernel spans 2r taps at stride r: exactly two inputs reach an output
    for (int64_t j = j_hi - 1; j <= j_hi; ++j) {
        if (j < 0 || j >= Lib) continue;
        const int k = (int)(q - j * r);             // in [0, 2r)
        for (int c = 0; c < C; ++c)
            acc += act(x[((int64_t)b * C + c) * Lin + j], 0.2f) * W[((int64_t)c * C + o) * (2 * r) + k];
    }
    y[i] = acc;
}

__global__ void g_add_inplace(float *x, const float *s, int64_t n, int64_t Lrow, int C, const int *lens, int spf)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (lens && (i % Lrow) >= (int64_t)lens[(int)(i / (Lrow * C))] * spf) return;
    x[i] += s[i];
}

// location_variable_convolution + gate + residual (modules.py:213-217,220-253; its dilation argument is always 1):
//   z[o][l*hop + s] = bias[layer][o][l] + sum_{c,k} ypad[c][l*hop + s + k - (ks-1)/2] * K[layer][c][o][k][l]     (zero pad of the WHOLE signal)
//   out = x + sigmoid(z[ch]) * tanh(z[ch + C]);   thread = (b, ch, t).  kernels [B][layers*C*2C*ks][T], biases [B][layers*2C][T].
__global__ void g_lvc_gate(const float *y, const float *kernels, const float *biases, const float *x, float *out, int B, int C, int ks,
                           int layers, int layer, int hop, int T, const int *lens)
{
    const int64_t Ln = (int64_t)T * hop;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C * Ln) return;
    const int64_t t = i % Ln;
    const int ch = (int)((i / Ln) % C), b = (int)(i / (Ln * C));
    const int64_t Lnb = lens ? (int64_t)lens[b] * hop : Ln;
    if (t >= Lnb) return;
    const int l = (int)(t / hop), half = (ks - 1) / 2;
    const float *kb = kernels + (int64_t)b * layers * C * 2 * C * ks * T;
    const float *bb = biases + (int64_t)b * layers * 2 * C * T;
    float zs = bb[((int64_t)layer * 2 * C + ch) * T + l], zt = bb[((int64_t)layer * 2 * C + ch + C) * T + l];
    for (int c = 0; c < C; ++c) {
        const float *yr = y + ((int64_t)b * C + c) * Ln;
        for (int k = 0; k < ks; ++k) {
            const int64_t p = t + k - half;
            if (p < 0 || p >= Lnb) continue;
            const float v = yr[p];
            const int64_t base = (((int64_t)layer * C + c) * 2 * C) * ks;
            zs += v * kb[(base + (int64_t)ch * ks + k) * T + l];
            zt += v * kb[(base + (int64_t)(ch + C) * ks + k) * T + l];
        }
    }
    out[i] = x[i] + (1.0f / (1.0f + expf(-zs))) * tanhf(zt);
}

// One reverse step of sampling_given_noise_schedule on x (util.py:219-229), scalar form of fdk::sampler_update4: the same Philox
// counters, so with a sample count per utterance that is a multiple of 4 the draws are the tuned path's.
__global__ void g_update(float *x, const float *eps, fd_step st, int k, int ddim, const float *z, unsigned long long seed,
                         const unsigned long long *uids, int64_t L, int64_t n, float *seq, const int *lens, int spf)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (lens && (i % L) >= (int64_t)lens[(int)(i / L)] * spf) return;
    const float xv = x[i], e = eps[i];
    float o;
    if (ddim) o = (st.c1 * xv + st.c2 * e) + st.c3 * e;
    else {
        o = (xv - st.c_eps * e) / st.c_div;
        if (st.add_noise) {
            float zv;
            if (z) zv = z[(int64_t)k * n + i];
            else {
                const int64_t b = i / L, t = i - b * L;
                const float4 q = uids ? fdk::philox_normal4(seed, (uint32_t)k, (uint64_t)(t >> 2), uids[b]) : fdk::philox_normal4(seed, (uint32_t)k, (uint64_t)(i >> 2));
                const int comp = (int)((uids ? t : i) & 3);
                zv = comp == 0 ? q.x : (comp == 1 ? q.y : (comp == 2 ? q.z : q.w));
            }
            o += st.sigma * zv;
        }
    }
    x[i] = o;
    if (seq) seq[(int64_t)(k + 1) * n + i] = o;
}

__global__ void g_init_noise(float *x, unsigned long long seed, const unsigned long long *uids, int64_t L, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / L, t = i - b * L;
    const float4 q = uids ? fdk::philox_normal4(seed, 0xFFFFFFFFu, (uint64_t)(t >> 2), uids[b]) : fdk::philox_normal4(seed, 0xFFFFFFFFu, (uint64_t)(i >> 2));
    const int comp = (int)((uids ? t : i) & 3);
    x[i] = comp == 0 ? q.x : (comp == 1 ? q.y : (comp == 2 ? q.z : q.w));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
static inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

#define G_LAUNCH(kern, n, ...) do { hipLaunchKernelGGL(kern, grid1(n), dim3(256), 0, stream, __VA_ARGS__); \
                                    hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

int validate(const fd_config &c, std::string &why)
{
    auto bad = [&](const char *m) { why = m; return FD_ERR_UNSUPPORTED; };
    if (c.audio_channels != 1) return bad("audio_channels != 1: first_audio_conv is Conv1d(1, inner_channels) in the reference too (FastDiff_model.py:34), its sampler feeds it one channel");
    if (c.inner_channels < 1 || c.cond_channels < 1 || c.kpnet_hidden_channels < 1) return bad("channel counts must be positive");
    if (c.n_upsample < 1 || c.n_upsample > 8) return bad("1..8 upsample stages");
    int64_t hop = 1;
    for (int n = 0; n < c.n_upsample; ++n) {
        // ratio 1: the reference builds ConvTranspose1d(stride = 1, output_padding = 1), which torch refuses (output padding must be
        // smaller than the stride, modules.py:163-166)
        if (c.upsample_ratios[n] < 2) return bad("upsample ratios must be >= 2 (the reference's ConvTranspose1d(stride = r, output_padding = r % 2) does not exist for r = 1)");
        hop *= c.upsample_ratios[n];
        if (hop > 65536) return bad("the product of the upsample ratios (samples per frame) must be <= 65536");
    }
    // the kernels index channels and per-frame coefficient blocks with int: bound every product they form
    if (c.inner_channels > 1024 || c.cond_channels > 4096 || c.kpnet_hidden_channels > 4096) return bad("inner_channels <= 1024, cond_channels <= 4096, kpnet_hidden_channels <= 4096");
    if (c.lvc_kernel_size > 31 || c.kpnet_conv_size > 31) return bad("kernel sizes <= 31");
    if (c.diffusion_step_embed_dim_in > 65536 || c.diffusion_step_embed_dim_mid > 65536 || c.diffusion_step_embed_dim_out > 65536) return bad("embedding widths <= 65536");
    if (c.lvc_layers_each_block < 1 || c.lvc_layers_each_block > 8) return bad("1..8 LVC layers per block (dilation 3^i)");
    if (c.lvc_kernel_size < 1 || (c.lvc_kernel_size & 1) == 0) return bad("an even lvc_kernel_size changes the sequence length: the reference's own forward fails on it (modules.py:183-187,217)");
    if (c.kpnet_conv_size < 1 || (c.kpnet_conv_size & 1) == 0) return bad("an even kpnet_conv_size changes the frame count: the reference's own forward fails on it (modules.py:293-318)");
    if (c.diffusion_step_embed_dim_in < 4 || (c.diffusion_step_embed_dim_in & 1)) return bad("diffusion_step_embed_dim_in must be even (util.py:423) and >= 4");
    if (c.diffusion_step_embed_dim_mid < 1 || c.diffusion_step_embed_dim_out < 1) return bad("embedding widths must be positive");
    if ((int64_t)c.lvc_layers_each_block * 2 * c.inner_channels * c.inner_channels * c.lvc_kernel_size > ((int64_t)1 << 24))
        return bad("layers * 2 * inner_channels^2 * lvc_kernel_size (kernel_conv's output channels) must be <= 2^24");
    return FD_OK;
}

int create(fd_context *c)
{
    Net *n = new Net();
    n->cfg = c->cfg;
    n->nb = c->cfg.n_upsample;
    int h = 1;
    for (int i = 0; i < n->nb; ++i) { h *= c->cfg.upsample_ratios[i]; n->hop[i] = h; n->blk[i].convs.resize(c->cfg.lvc_layers_each_block); }
    n->hop_total = h;
    c->gen = n;
    return FD_OK;
}

static void free_weights(Net *n)
{
    for (void *p : n->allocs) (void)hipFree(p);
    n->allocs.clear();
}

void destroy(fd_context *c)
{
    if (!c->gen) return;
    free_weights(c->gen);
    if (c->gen->ws) (void)hipFree(c->gen->ws);
    delete c->gen;
    c->gen = nullptr;
}

int hop_total(const fd_context *c) { return c->gen ? c->gen->hop_total : fd::HOPT; }

static int up(fd_context *c, Net *n, const std::vector<float> &v, const float **dst)
{
    void *d = nullptr;
    if (hipMalloc(&d, sizeof(float) * std::max<size_t>(v.size(), 1)) != hipSuccess) { c->err = "fd_commit_weights: out of device memory"; return FD_ERR_HIP; }
    n->allocs.push_back(d);
    if (hipMemcpy(d, v.data(), sizeof(float) * v.size(), hipMemcpyHostToDevice) != hipSuccess) { c->err = "fd_commit_weights: upload failed"; return FD_ERR_HIP; }
    *dst = static_cast<const float *>(d);
    return FD_OK;
}

int commit(fd_context *c, const std::map<std::string, FoldedParam> &f)
{
    Net *n = c->gen;
    (void)hipDeviceSynchronize();
    free_weights(n);
    int rc;
    auto conv = [&](const std::string &name, Conv &cw) -> int {
        const auto it = f.find(name);
        if (it == f.end()) { c->err = "fd_commit_weights: missing parameter " + name; return FD_ERR_MISSING; }
        if ((rc = up(c, n, it->second.w, &cw.w)) != FD_OK) return rc;
        return up(c, n, it->second.b, &cw.b);
    };
    static const int KP_RES_IDX[6] = {1, 3, 6, 8, 11, 13};
    if ((rc = conv("first_audio_conv", n->first)) != FD_OK) return rc;
    if ((rc = conv("final_conv.0", n->final_)) != FD_OK) return rc;
    if ((rc = conv("fc_t1", n->fc_t1)) != FD_OK) return rc;
    if ((rc = conv("fc_t2", n->fc_t2)) != FD_OK) return rc;
    for (int b = 0; b < n->nb; ++b) {
        const std::string p = "lvc_blocks." + std::to_string(b), d = "downsample." + std::to_string(b);
        Net::Blk &k = n->blk[b];
        if ((rc = conv(p + ".fc_t", k.fc_t)) != FD_OK) return rc;
        if ((rc = conv(p + ".upsample", k.up)) != FD_OK) return rc;
        if ((rc = conv(p + ".kernel_predictor.input_conv.0", k.kp_in)) != FD_OK) return rc;
        for (int j = 0; j < 6; ++j)
            if ((rc = conv(p + ".kernel_predictor.residual_conv." + std::to_string(KP_RES_IDX[j]), k.kp_res[j])) != FD_OK) return rc;
        if ((rc = conv(p + ".kernel_predictor.kernel_conv", k.kc)) != FD_OK) return rc;
        if ((rc = conv(p + ".kernel_predictor.bias_conv", k.bc)) != FD_OK) return rc;
        for (size_t i = 0; i < k.convs.size(); ++i)
            if ((rc = conv(p + ".convs." + std::to_string(i), k.convs[i])) != FD_OK) return rc;
        if ((rc = conv(d + ".residual_dense", k.res)) != FD_OK) return rc;
        for (int i = 0; i < 3; ++i)
            if ((rc = conv(d + ".conv." + std::to_string(i), k.dconv[i])) != FD_OK) return rc;
    }
    {   // frequency table of calc_diffusion_step_embedding (util.py:425-427): fp32 product, fp32 exp
        const int half = n->cfg.diffusion_step_embed_dim_in / 2;
        std::vector<float> table(half);
        const float cst = (float)(-(log(10000.0) / (double)(half - 1)));
        for (int j = 0; j < half; ++j) {
            volatile float arg = (float)j * cst;
            table[j] = expf(arg);
        }
        if ((rc = up(c, n, table, &n->embed_table)) != FD_OK) return rc;
    }
    return FD_OK;
}

// workspace layout of one call (floats)
struct Plan {
    int64_t L, T;
    size_t emb, mid, eout, noise, a[9], h0, h1, res, kph[3], kc, bc, xa, xb, y, eps, xs, uid, lens, total;
};

static Plan plan(const Net *n, int B, int T)
{
    Plan p;
    const fd_config &c = n->cfg;
    p.T = T; p.L = (int64_t)T * n->hop_total;
    size_t off = 0;
    auto take = [&](size_t floats) { const size_t o = off; off += (floats + 3) & ~(size_t)3; return o; };
    const size_t C = c.inner_channels;
    p.emb = take((size_t)B * c.diffusion_step_embed_dim_in);
    p.mid = take((size_t)B * c.diffusion_step_embed_dim_mid);
    p.eout = take((size_t)B * c.diffusion_step_embed_dim_out);
    p.noise = take((size_t)n->nb * B * c.cond_channels);
    int64_t len = p.L;
    for (int d = 0; d <= n->nb; ++d) {
        p.a[d] = take((size_t)B * C * len);
        if (d < n->nb) len /= c.upsample_ratios[n->nb - 1 - d];
    }
    const int64_t l1 = p.L / c.upsample_ratios[n->nb - 1];      // the longest DBlock output
    p.h0 = take((size_t)B * C * l1); p.h1 = take((size_t)B * C * l1); p.res = take((size_t)B * C * l1);
    for (int i = 0; i < 3; ++i) p.kph[i] = take((size_t)B * c.kpnet_hidden_channels * T);
    p.kc = take((size_t)B * c.lvc_layers_each_block * C * 2 * C * c.lvc_kernel_size * T);
    p.bc = take((size_t)B * c.lvc_layers_each_block * 2 * C * T);
    p.xa = take((size_t)B * C * p.L); p.xb = take((size_t)B * C * p.L); p.y = take((size_t)B * C * p.L);
    p.eps = take((size_t)B * p.L); p.xs = take((size_t)B * p.L);
    p.uid = take((size_t)2 * B);
    p.lens = take((size_t)B);
    p.total = off;
    return p;
}

static int ensure_ws(fd_context *c, const Plan &p)
{
    Net *n = c->gen;
    if (n->ws_floats >= p.total) return FD_OK;
    (void)hipDeviceSynchronize();
    if (n->ws) (void)hipFree(n->ws);
    n->ws = nullptr; n->ws_floats = 0;
    if (hipMalloc(reinterpret_cast<void **>(&n->ws), sizeof(float) * p.total) != hipSuccess) {
        c->err = "fastdiff_hip: out of device memory for the generic path's workspace (" + std::to_string(p.total * 4 >> 20) + " MiB)";
        return FD_ERR_HIP;
    }
    n->ws_floats = p.total;
    return FD_OK;
}

// eps = net((x, mel, t)): FastDiff.forward, FastDiff_model.py:74-102.  steps [B] device, or NULL with one value t_all for every utterance.
static hipError_t forward_dev(Net *n, const Plan &p, const float *x, const float *mel, const float *steps, float t_all, int B, float *eps_out,
                              const int *lens, hipStream_t stream)
{
    const fd_config &c = n->cfg;
    float *w = n->ws;
    const int C = c.inner_channels, CC = c.cond_channels, HID = c.kpnet_hidden_channels, KS = c.lvc_kernel_size, KK = c.kpnet_conv_size;
    const int LY = c.lvc_layers_each_block, T = (int)p.T;
    const int E_IN = c.diffusion_step_embed_dim_in, E_MID = c.diffusion_step_embed_dim_mid, E_OUT = c.diffusion_step_embed_dim_out;
    const float *none = nullptr;
    // a1, a2: embedding, two swish layers, the per-block fc_t
    G_LAUNCH(g_step_embed, (int64_t)B * (E_IN / 2), steps, t_all, n->embed_table, E_IN / 2, w + p.emb, B);
    G_LAUNCH(g_linear, (int64_t)B * E_MID, (const float *)(w + p.emb), n->fc_t1.w, n->fc_t1.b, w + p.mid, B, E_IN, E_MID, 1);
    G_LAUNCH(g_linear, (int64_t)B * E_OUT, (const float *)(w + p.mid), n->fc_t2.w, n->fc_t2.b, w + p.eout, B, E_MID, E_OUT, 1);
    for (int b = 0; b < n->nb; ++b)
        G_LAUNCH(g_linear, (int64_t)B * CC, (const float *)(w + p.eout), n->blk[b].fc_t.w, n->blk[b].fc_t.b, w + p.noise + (size_t)b * B * CC, B, E_OUT, CC, 0);
    // a3: first_audio_conv
    G_LAUNCH(g_conv1d, (int64_t)B * C * p.L, x, n->first.w, n->first.b, w + p.a[0], B, 1, C, 7, 1, 3, p.L, p.L, 1, none, 1.0f, 1.0f, none, (int64_t)0, 1, lens, n->hop_total);
    // a4: the DBlocks, factors = the ratios reversed (FastDiff_model.py:63); block d consumes a[d], leaves a[d + 1]
    int64_t len = p.L;
    for (int d = 0; d < n->nb; ++d) {
        const int f = c.upsample_ratios[n->nb - 1 - d];
        const int64_t lo = len / f;
        const int spf = (int)(lo / T);                // samples per frame at this DBlock's output rate
        const Net::Blk &k = n->blk[d];
        const float *src = w + p.a[d];
        // res = Conv1x1(x) picked at every f-th sample; h = the picked x through three (lrelu 0.2, conv k3 dilation 1, 2, 4); out = h + res
        G_LAUNCH(g_conv1d, (int64_t)B * C * lo, src, k.res.w, k.res.b, w + p.res, B, C, C, 1, 1, 0, len, lo, f, none, 1.0f, 1.0f, none, (int64_t)0, 1, lens, spf);
        G_LAUNCH(g_conv1d, (int64_t)B * C * lo, src, k.dconv[0].w, k.dconv[0].b, w + p.h0, B, C, C, 3, 1, 1, len, lo, f, none, 0.2f, 1.0f, none, (int64_t)0, 1, lens, spf);
        G_LAUNCH(g_conv1d, (int64_t)B * C * lo, (const float *)(w + p.h0), k.dconv[1].w, k.dconv[1].b, w + p.h1, B, C, C, 3, 2, 2, lo, lo, 1, none, 0.2f, 1.0f, none, (int64_t)0, 1, lens, spf);
        G_LAUNCH(g_conv1d, (int64_t)B * C * lo, (const float *)(w + p.h1), k.dconv[2].w, k.dconv[2].b, w + p.a[d + 1], B, C, C, 3, 4, 4, lo, lo, 1, none, 0.2f, 1.0f,
                 (const float *)(w + p.res), lo, 1, lens, spf);
        len = lo;
    }
    // the LVC blocks (modules.py:189-218): x starts as the bottom of the down path
    const float *xcur = w + p.a[n->nb];
    int64_t lin = T;
    for (int b = 0; b < n->nb; ++b) {
        const Net::Blk &k = n->blk[b];
        const int r = c.upsample_ratios[b], hop = n->hop[b];
        const int64_t ln = lin * r;
        const float *skip = w + p.a[n->nb - 1 - b];
        const float *nz = w + p.noise + (size_t)b * B * CC;
        // a5: KernelPredictor on c + noise: input conv k5 + lrelu 0.1; h + six (conv, lrelu 0.1); kernel_conv, bias_conv
        float *h0 = w + p.kph[0], *ha = w + p.kph[1], *hb = w + p.kph[2];
        G_LAUNCH(g_conv1d, (int64_t)B * HID * T, mel, k.kp_in.w, k.kp_in.b, h0, B, CC, HID, 5, 1, 2, (int64_t)T, (int64_t)T, 1, nz, 1.0f, 0.1f, none, (int64_t)0, 1, lens, 1);
        const float *cur = h0;
        for (int j = 0; j < 6; ++j) {
            float *dst = (j & 1) ? hb : ha;
            const bool last = j == 5;
            G_LAUNCH(g_conv1d, (int64_t)B * HID * T, cur, k.kp_res[j].w, k.kp_res[j].b, dst, B, HID, HID, KK, 1, (KK - 1) / 2, (int64_t)T, (int64_t)T, 1, none, 1.0f, 0.1f,
                     last ? (const float *)h0 : none, (int64_t)T, 1, lens, 1);
            cur = dst;
        }
        G_LAUNCH(g_conv1d, (int64_t)B * LY * C * 2 * C * KS * T, cur, k.kc.w, k.kc.b, w + p.kc, B, HID, LY * C * 2 * C * KS, KK, 1, (KK - 1) / 2, (int64_t)T, (int64_t)T, 1, none,
                 1.0f, 1.0f, none, (int64_t)0, 1, lens, 1);
        G_LAUNCH(g_conv1d, (int64_t)B * LY * 2 * C * T, cur, k.bc.w, k.bc.b, w + p.bc, B, HID, LY * 2 * C, KK, 1, (KK - 1) / 2, (int64_t)T, (int64_t)T, 1, none, 1.0f, 1.0f,
                 none, (int64_t)0, 1, lens, 1);
        // a6: x = upsample(lrelu(x, 0.2)) into the ping-pong buffer the input does not occupy
        float *dst = (xcur == w + p.xa) ? w + p.xb : w + p.xa;
        float *oth = (dst == w + p.xa) ? w + p.xb : w + p.xa;
        G_LAUNCH(g_convt, (int64_t)B * C * ln, xcur, k.up.w, k.up.b, dst, B, C, r, r / 2 + r % 2, lin, lens, (int)(lin / T));
        // a7-a9: per layer x += skip; y = lrelu(conv_{ks, dilation 3^i}(lrelu(x))); x = x + gate(LVC(y))
        int dil = 1;
        for (int i = 0; i < LY; ++i) {
            G_LAUNCH(g_add_inplace, (int64_t)B * C * ln, dst, skip, (int64_t)B * C * ln, ln, C, lens, hop);
            G_LAUNCH(g_conv1d, (int64_t)B * C * ln, (const float *)dst, k.convs[i].w, k.convs[i].b, w + p.y, B, C, C, KS, dil, dil * ((KS - 1) / 2), ln, ln, 1, none, 0.2f, 0.2f,
                     none, (int64_t)0, 1, lens, hop);
            G_LAUNCH(g_lvc_gate, (int64_t)B * C * ln, (const float *)(w + p.y), (const float *)(w + p.kc), (const float *)(w + p.bc), (const float *)dst, oth, B, C, KS, LY, i, hop, T, lens);
            std::swap(dst, oth);
            dil *= 3;
        }
        xcur = dst;
        lin = ln;
    }
    // a10: final_conv
    G_LAUNCH(g_conv1d, (int64_t)B * p.L, xcur, n->final_.w, n->final_.b, eps_out, B, C, 1, 7, 1, 3, p.L, p.L, 1, none, 1.0f, 1.0f, none, (int64_t)0, 1, lens, n->hop_total);
    return hipSuccess;
}

// the caller's `lens` (host, nullable) -> the workspace's device copy, or NULL when every utterance fills the batch
static const int *upload_lens(Net *n, const Plan &p, const int *lens, int B, int T, hipStream_t stream, hipError_t *err)
{
    *err = hipSuccess;
    if (!lens) return nullptr;
    bool ragged = false;
    for (int b = 0; b < B; ++b) ragged = ragged || lens[b] < T;
    if (!ragged) return nullptr;
    int *d = reinterpret_cast<int *>(n->ws + p.lens);
    *err = hipMemcpyAsync(d, lens, sizeof(int) * B, hipMemcpyHostToDevice, stream);
    if (*err == hipSuccess) *err = hipStreamSynchronize(stream);      // `lens` is the caller's array: it may be reused as soon as we return
    return d;
}

int forward(fd_context *c, const float *x, const float *mel, const float *steps, int B, int T, const int *lens, float *eps_out, hipStream_t stream)
{
    Net *n = c->gen;
    const Plan p = plan(n, B, T);
    int rc = ensure_ws(c, p);
    if (rc != FD_OK) return rc;
    hipError_t e = hipSuccess;
    const int *dl = upload_lens(n, p, lens, B, T, stream, &e);
    if (e == hipSuccess) e = forward_dev(n, p, x, mel, steps, 0.0f, B, eps_out, dl, stream);
    if (e != hipSuccess) { c->err = std::string("fd_forward (generic configuration): ") + hipGetErrorString(e); return FD_ERR_HIP; }
    return FD_OK;
}

int sample(fd_context *c, const float *mel, int B, int T, const int *lens, const fd_step *table, int N, int ddim, const float *x_T, const float *z,
           unsigned long long seed, const std::vector<unsigned long long> &ids, float *out, float *seq_out, hipStream_t stream)
{
    Net *n = c->gen;
    const Plan p = plan(n, B, T);
    int rc = ensure_ws(c, p);
    if (rc != FD_OK) return rc;
    const int64_t cnt = (int64_t)B * p.L;
    float *xs = n->ws + p.xs, *eps = n->ws + p.eps;
    unsigned long long *uids = nullptr;
    hipError_t e = hipSuccess;
    const int *dl = upload_lens(n, p, lens, B, T, stream, &e);
    if (e == hipSuccess && !ids.empty()) {
        uids = reinterpret_cast<unsigned long long *>(n->ws + p.uid);
        e = hipMemcpyAsync(uids, ids.data(), sizeof(unsigned long long) * B, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);      // `ids` is the caller's vector: gone when we return
    }
    auto run = [&]() -> hipError_t {
        if (x_T) { if (hipMemcpyAsync(xs, x_T, sizeof(float) * cnt, hipMemcpyDeviceToDevice, stream) != hipSuccess) return hipGetLastError(); }
        else G_LAUNCH(g_init_noise, cnt, xs, seed, (const unsigned long long *)uids, p.L, cnt);
        if (seq_out && hipMemcpyAsync(seq_out, xs, sizeof(float) * cnt, hipMemcpyDeviceToDevice, stream) != hipSuccess) return hipGetLastError();
        for (int k = 0; k < N; ++k) {
            const hipError_t ef = forward_dev(n, p, xs, mel, nullptr, table[k].t, B, eps, dl, stream);
            if (ef != hipSuccess) return ef;
            G_LAUNCH(g_update, cnt, xs, (const float *)eps, table[k], k, ddim, z, seed, (const unsigned long long *)uids, p.L, cnt, seq_out, dl, n->hop_total);
        }
        if (hipMemcpyAsync(out, xs, sizeof(float) * cnt, hipMemcpyDeviceToDevice, stream) != hipSuccess) return hipGetLastError();
        return hipSuccess;
    };
    if (e == hipSuccess) e = run();
    if (e != hipSuccess) { c->err = std::string("fd_sample (generic configuration): ") + hipGetErrorString(e); return FD_ERR_HIP; }
    return FD_OK;
}

}  // namespace fdg
