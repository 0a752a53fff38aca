/*
 * fastdiff_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the FastDiff vocoder inference path of the
 * reference (Rongjiehuang/FastDiff): the noise-predictor denoiser forward and
 * the N-step reverse sampling loop.  It exists only so that tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() can check the HIP path; nothing
 * under fastdiff_amd/ may import, link or call it.
 *
 * Pinned against the reference itself: tests/golden/ holds tensors produced by
 * importing the reference's own PyTorch modules (oracle/gen_golden.py), and
 * tests/test_oracle_golden.py checks every function below against them.
 *
 * Built twice by oracle/build.py:  -DFD_REAL=double -> libfdoracle_f64.so
 *                                  -DFD_REAL=float  -> libfdoracle_f32.so
 * All tensors are [B][C][time], time innermost, zero padding everywhere.
 *
 * Citations are path:line in the reference repository.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifndef FD_REAL
#define FD_REAL double
#endif
typedef FD_REAL real;

#define FDO_MAX_BLOCKS 8

/* Architecture hyper-parameters: FastDiff.__init__ (modules/FastDiff/module/FastDiff_model.py:13-26) */
typedef struct {
    int inner_channels;   /* 32 */
    int cond_channels;    /* 80 */
    int n_blocks;         /* 3  */
    int ratios[FDO_MAX_BLOCKS]; /* upsample_ratios [8,8,4] */
    int lvc_layers;       /* 4  */
    int lvc_kernel_size;  /* 3  */
    int kp_hidden;        /* 64 */
    int kp_conv_size;     /* 3  */
    int embed_in;         /* 128 */
    int embed_mid;        /* 512 */
    int embed_out;        /* 512 */
} fdo_config;

int fdo_real_bytes(void) { return (int)sizeof(real); }

#ifdef _OPENMP
#include <omp.h>
int fdo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int fdo_set_threads(int n) { (void)n; return 1; }
#endif

/* ------------------------------------------------------------------ */
/* elementary ops                                                      */
/* ------------------------------------------------------------------ */

/* F.leaky_relu (modules.py:135,205,210,212) */
void fdo_leaky_relu(real *x, int64_t n, double slope)
{
    const real s = (real)slope;
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] > 0 ? x[i] : x[i] * s;
}

/* torch.nn.utils.weight_norm fold, dim=0: w[o] = v[o] * (g[o] / ||v[o]||_2)
 * (FastDiff_model.py:115-122; torch._weight_norm(v, g, 0)) */
void fdo_weight_norm_fold(const real *v, const real *g, int cout, int per_out, real *w)
{
    for (int o = 0; o < cout; ++o) {
        real ss = 0;
        for (int j = 0; j < per_out; ++j) ss += v[(int64_t)o * per_out + j] * v[(int64_t)o * per_out + j];
        real scale = g[o] / (real)sqrt((double)ss);
        for (int j = 0; j < per_out; ++j) w[(int64_t)o * per_out + j] = v[(int64_t)o * per_out + j] * scale;
    }
}

/* torch.nn.Conv1d, stride 1, zero padding pad = dil*(ks-1)/2 (length preserving), cross-correlation.
 * out[b,o,t] = bias[o] + sum_i sum_k w[o,i,k] * x[b,i,t + k*dil - pad]
 * (FastDiff_model.py:34-36,67-68; modules.py:120-125,185,293-318) */
void fdo_conv1d(const real *x, int B, int cin, int64_t L, const real *w, const real *bias,
                int cout, int ks, int dil, real *out)
{
    const int pad = dil * (ks - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < cout; ++o) {
            real *orow = out + ((int64_t)b * cout + o) * L;
            const real bv = bias ? bias[o] : (real)0;
            for (int64_t t = 0; t < L; ++t) orow[t] = bv;
            for (int i = 0; i < cin; ++i) {
                const real *xrow = x + ((int64_t)b * cin + i) * L;
                for (int k = 0; k < ks; ++k) {
                    const real wv = w[((int64_t)o * cin + i) * ks + k];
                    const int64_t sh = (int64_t)k * dil - pad;
                    int64_t t0 = sh < 0 ? -sh : 0;
                    int64_t t1 = sh > 0 ? L - sh : L;
                    for (int64_t t = t0; t < t1; ++t) orow[t] += wv * xrow[t + sh];
                }
            }
        }
}

/* torch.nn.ConvTranspose1d(C, C, 2r, stride=r, padding=r//2 + r%2, output_padding=r%2), weight [in,out,k]
 * out[b,o,t] = bias[o] + sum_i sum_j x[b,i,j] * w[i,o,t + p - j*r]   (modules.py:163-166,206) */
void fdo_conv_transpose1d(const real *x, int B, int cin, int64_t Lin, const real *w, const real *bias,
                          int cout, int r, real *out)
{
    const int ks = 2 * r, p = r / 2 + r % 2, op = r % 2;
    const int64_t Lout = (Lin - 1) * r - 2 * p + ks + op;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < cout; ++o) {
            real *orow = out + ((int64_t)b * cout + o) * Lout;
            for (int64_t t = 0; t < Lout; ++t) orow[t] = bias[o];
            for (int i = 0; i < cin; ++i) {
                const real *xrow = x + ((int64_t)b * cin + i) * Lin;
                const real *wk = w + ((int64_t)i * cout + o) * ks;
                for (int64_t j = 0; j < Lin; ++j) {
                    const real xv = xrow[j];
                    for (int k = 0; k < ks; ++k) {
                        int64_t t = j * r + k - p;
                        if (t >= 0 && t < Lout) orow[t] += xv * wk[k];
                    }
                }
            }
        }
}

/* torch.nn.Linear: y[b,o] = bias[o] + sum_i w[o,i] x[b,i] */
void fdo_linear(const real *x, int B, int nin, const real *w, const real *bias, int nout, real *y)
{
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < nout; ++o) {
            real acc = bias[o];
            for (int i = 0; i < nin; ++i) acc += w[(int64_t)o * nin + i] * x[(int64_t)b * nin + i];
            y[(int64_t)b * nout + o] = acc;
        }
}

static inline real fdo_sigmoid(real x) { return (real)1 / ((real)1 + (real)exp((double)-x)); }

/* swish (FastDiff_model.py:7-8) */
void fdo_swish(real *x, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] * fdo_sigmoid(x[i]);
}

/* ------------------------------------------------------------------ */
/* a1: calc_diffusion_step_embedding (util.py:407-432)                 */
/* ------------------------------------------------------------------ */

/* The frequency table is float32 in the reference whatever the model dtype:
 * torch.exp(torch.arange(half) * -(np.log(10000)/(half-1))) -> fp32 product, fp32 exp. */
void fdo_embed_table(int half_dim, float *table)
{
    const float c = (float)(-(log(10000.0) / (double)(half_dim - 1)));
    for (int j = 0; j < half_dim; ++j) {
        volatile float arg = (float)j * c;   /* fp32 product, as the int64*scalar TensorIterator op computes it */
        table[j] = expf(arg);
    }
}

/* steps [B] (fractional at inference, util.py:217), table [half] -> emb [B, 2*half] = cat(sin, cos) */
void fdo_step_embedding(const real *steps, int B, const float *table, int half_dim, real *emb)
{
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < half_dim; ++j) {
            real arg = steps[b] * (real)table[j];
            if (sizeof(real) == 4) {
                emb[(int64_t)b * 2 * half_dim + j] = (real)sinf((float)arg);
                emb[(int64_t)b * 2 * half_dim + half_dim + j] = (real)cosf((float)arg);
            } else {
                emb[(int64_t)b * 2 * half_dim + j] = (real)sin((double)arg);
                emb[(int64_t)b * 2 * half_dim + half_dim + j] = (real)cos((double)arg);
            }
        }
}

/* ------------------------------------------------------------------ */
/* a4: DiffusionDBlock (modules.py:116-138)                            */
/* ------------------------------------------------------------------ */
/* F.interpolate(x, size=L//f) (nearest) with integer factor == x[..., ::f].
 * w: res_w [C,C,1], res_b, then 3x (conv_w [C,C,3], conv_b), dilations 1,2,4. */
void fdo_dblock(const real *x, int B, int C, int64_t Lin, int factor, const real *const *w, real *out)
{
    const int64_t Lo = Lin / factor;
    real *res_full = (real *)malloc(sizeof(real) * (size_t)B * C * Lin);
    real *h = (real *)malloc(sizeof(real) * (size_t)B * C * Lo);
    real *h2 = (real *)malloc(sizeof(real) * (size_t)B * C * Lo);
    /* residual = interpolate(residual_dense(x)) : 1x1 conv at full rate, then strided pick (:130-131) */
    fdo_conv1d(x, B, C, Lin, w[0], w[1], C, 1, 1, res_full);
    /* x = interpolate(x) (:133) */
    for (int64_t bc = 0; bc < (int64_t)B * C; ++bc)
        for (int64_t t = 0; t < Lo; ++t) h[bc * Lo + t] = x[bc * Lin + t * factor];
    const int dil[3] = {1, 2, 4};
    for (int l = 0; l < 3; ++l) {   /* :134-136 */
        fdo_leaky_relu(h, (int64_t)B * C * Lo, 0.2);
        fdo_conv1d(h, B, C, Lo, w[2 + 2 * l], w[3 + 2 * l], C, 3, dil[l], h2);
        real *tmp = h; h = h2; h2 = tmp;
    }
    for (int64_t bc = 0; bc < (int64_t)B * C; ++bc)   /* :138 */
        for (int64_t t = 0; t < Lo; ++t) out[bc * Lo + t] = h[bc * Lo + t] + res_full[bc * Lin + t * factor];
    free(res_full); free(h); free(h2);
}

/* ------------------------------------------------------------------ */
/* a5: KernelPredictor (modules.py:257-343)                            */
/* ------------------------------------------------------------------ */
/* cond [B,cond,T]; w: in_w [H,cond,5], in_b, 6x(res_w [H,H,ks], res_b), kc_w [l_w,H,ks], kc_b, bc_w [l_b,H,ks], bc_b
 * kernels out: [B, l_w, T] (== view [B,layers,in,out,k,T]); bias out: [B, l_b, T] (== [B,layers,out,T]) */
void fdo_kernel_predictor(const real *cond, int B, int ccond, int T, int H, int ks, int l_w, int l_b,
                          const real *const *w, real *kernels, real *bias)
{
    const int64_t n = (int64_t)B * H * T;
    real *h = (real *)malloc(sizeof(real) * (size_t)n);
    real *r = (real *)malloc(sizeof(real) * (size_t)n);
    real *r2 = (real *)malloc(sizeof(real) * (size_t)n);
    fdo_conv1d(cond, B, ccond, T, w[0], w[1], H, 5, 1, h);   /* input_conv (:293-296) */
    fdo_leaky_relu(h, n, 0.1);
    memcpy(r, h, sizeof(real) * (size_t)n);
    for (int l = 0; l < 6; ++l) {                            /* residual_conv (:298-313), dropout p=0 */
        fdo_conv1d(r, B, H, T, w[2 + 2 * l], w[3 + 2 * l], H, ks, 1, r2);
        fdo_leaky_relu(r2, n, 0.1);
        real *tmp = r; r = r2; r2 = tmp;
    }
    for (int64_t i = 0; i < n; ++i) h[i] += r[i];            /* c = c + residual_conv(c) (:329) */
    fdo_conv1d(h, B, H, T, w[14], w[15], l_w, ks, 1, kernels);  /* kernel_conv (:330) */
    fdo_conv1d(h, B, H, T, w[16], w[17], l_b, ks, 1, bias);     /* bias_conv (:331) */
    free(h); free(r); free(r2);
}

/* ------------------------------------------------------------------ */
/* a8: location_variable_convolution (modules.py:220-253), dilation 1  */
/* ------------------------------------------------------------------ */
/* x [B,cin,L], kernel [B,cin,cout,ks,T] (T innermost), bias [B,cout,T], L == T*hop
 * out[b,o,l*hop+s] = bias[b,o,l] + sum_{i,k} xpad[b,i,l*hop+s+k] * kernel[b,i,o,k,l],
 * xpad = x zero-padded by (ks-1)/2 at both ends of the whole signal. */
void fdo_lvc(const real *x, int B, int cin, int T, int hop, const real *kernel, const real *bias,
             int cout, int ks, real *out)
{
    const int64_t L = (int64_t)T * hop;
    const int pad = (ks - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < cout; ++o) {
            real *orow = out + ((int64_t)b * cout + o) * L;
            for (int l = 0; l < T; ++l) {
                const real bv = bias[((int64_t)b * cout + o) * T + l];
                for (int s = 0; s < hop; ++s) orow[(int64_t)l * hop + s] = bv;
                for (int i = 0; i < cin; ++i) {
                    const real *xrow = x + ((int64_t)b * cin + i) * L;
                    for (int k = 0; k < ks; ++k) {
                        const real kv = kernel[((((int64_t)b * cin + i) * cout + o) * ks + k) * T + l];
                        for (int s = 0; s < hop; ++s) {
                            int64_t p = (int64_t)l * hop + s + k - pad;
                            if (p >= 0 && p < L) orow[(int64_t)l * hop + s] += kv * xrow[p];
                        }
                    }
                }
            }
        }
}

/* ------------------------------------------------------------------ */
/* TimeAware_LVCBlock.forward (modules.py:190-218)                     */
/* ------------------------------------------------------------------ */
/* w (per block): fc_t_w [cond,eout], fc_t_b, up_w [C,C,2r], up_b, 18 KernelPredictor arrays, layers x (conv_w, conv_b)
 * x [B,C,Lin], skip [B,C,Lin*r], c [B,cond,T], embed [B,eout] -> out [B,C,Lin*r]
 * Optional taps (may be NULL): tap_kernels [B,l_w,T], tap_bias [B,l_b,T]. */
void fdo_lvc_block(const fdo_config *cfg, int blk, const real *x, const real *skip, const real *c,
                   const real *embed, int B, int T, int64_t Lin, int hop, const real *const *w,
                   real *out, real *tap_kernels, real *tap_bias)
{
    const int C = cfg->inner_channels, cc = cfg->cond_channels, r = cfg->ratios[blk];
    const int layers = cfg->lvc_layers, ks = cfg->lvc_kernel_size;
    const int l_w = C * 2 * C * ks * layers, l_b = 2 * C * layers;
    const int64_t Lo = Lin * r;
    real *noise = (real *)malloc(sizeof(real) * (size_t)B * cc);
    real *cond = (real *)malloc(sizeof(real) * (size_t)B * cc * T);
    real *kern = (real *)malloc(sizeof(real) * (size_t)B * l_w * T);
    real *kbias = (real *)malloc(sizeof(real) * (size_t)B * l_b * T);
    real *xin = (real *)malloc(sizeof(real) * (size_t)B * C * Lin);
    real *y = (real *)malloc(sizeof(real) * (size_t)B * C * Lo);
    real *y2 = (real *)malloc(sizeof(real) * (size_t)B * C * Lo);
    real *z = (real *)malloc(sizeof(real) * (size_t)B * 2 * C * Lo);
    real *lk = (real *)malloc(sizeof(real) * (size_t)B * C * 2 * C * ks * T);
    real *lb = (real *)malloc(sizeof(real) * (size_t)B * 2 * C * T);

    /* noise = fc_t(embed).unsqueeze(-1); condition = c + noise (:202-203) */
    fdo_linear(embed, B, cfg->embed_out, w[0], w[1], cc, noise);
    for (int b = 0; b < B; ++b)
        for (int ch = 0; ch < cc; ++ch)
            for (int t = 0; t < T; ++t)
                cond[((int64_t)b * cc + ch) * T + t] = c[((int64_t)b * cc + ch) * T + t] + noise[b * cc + ch];
    fdo_kernel_predictor(cond, B, cc, T, cfg->kp_hidden, cfg->kp_conv_size, l_w, l_b, w + 4, kern, kbias); /* :204 */
    if (tap_kernels) memcpy(tap_kernels, kern, sizeof(real) * (size_t)B * l_w * T);
    if (tap_bias) memcpy(tap_bias, kbias, sizeof(real) * (size_t)B * l_b * T);

    memcpy(xin, x, sizeof(real) * (size_t)B * C * Lin);
    fdo_leaky_relu(xin, (int64_t)B * C * Lin, 0.2);                       /* :205 */
    fdo_conv_transpose1d(xin, B, C, Lin, w[2], w[3], C, r, out);          /* :206 */

    const real *const *wc = w + 4 + 18;
    int dil = 1;
    for (int i = 0; i < layers; ++i, dil *= 3) {
        const int64_t n = (int64_t)B * C * Lo;
        for (int64_t j = 0; j < n; ++j) out[j] += skip[j];                /* x += audio_down, every layer (:209) */
        memcpy(y, out, sizeof(real) * (size_t)n);
        fdo_leaky_relu(y, n, 0.2);                                        /* :210 */
        fdo_conv1d(y, B, C, Lo, wc[2 * i], wc[2 * i + 1], C, ks, dil, y2); /* :211, dilation 3^i */
        fdo_leaky_relu(y2, n, 0.2);                                       /* :212 */
        /* k = kernels[:, i], b = bias[:, i] with kernels viewed [B,layers,in,out,k,T] (:214-215,333-342) */
        const int64_t per_layer_k = (int64_t)C * 2 * C * ks * T, per_layer_b = (int64_t)2 * C * T;
        for (int b = 0; b < B; ++b) {
            memcpy(lk + (int64_t)b * per_layer_k, kern + ((int64_t)b * layers + i) * per_layer_k,
                   sizeof(real) * (size_t)per_layer_k);
            memcpy(lb + (int64_t)b * per_layer_b, kbias + ((int64_t)b * layers + i) * per_layer_b,
                   sizeof(real) * (size_t)per_layer_b);
        }
        fdo_lvc(y2, B, C, T, hop, lk, lb, 2 * C, ks, z);                  /* :216 */
        /* x = x + sigmoid(y[:, :C]) * tanh(y[:, C:]) (:217) */
        for (int b = 0; b < B; ++b)
            for (int ch = 0; ch < C; ++ch)
                for (int64_t t = 0; t < Lo; ++t) {
                    real zs = z[((int64_t)b * 2 * C + ch) * Lo + t];
                    real zt = z[((int64_t)b * 2 * C + C + ch) * Lo + t];
                    out[((int64_t)b * C + ch) * Lo + t] += fdo_sigmoid(zs) * (real)tanh((double)zt);
                }
    }
    free(noise); free(cond); free(kern); free(kbias); free(xin); free(y); free(y2); free(z); free(lk); free(lb);
}

/* ------------------------------------------------------------------ */
/* FastDiff.forward (FastDiff_model.py:74-102)                         */
/* ------------------------------------------------------------------ */
/* Canonical folded-weight pointer order (w[]):
 *   0 first_w [C,1,7]  1 first_b  2 fc_t1_w [mid,in]  3 fc_t1_b  4 fc_t2_w [out,mid]  5 fc_t2_b
 *   then per downsample d (8 arrays): res_w [C,C,1], res_b, conv{0,1,2}_w [C,C,3], conv{0,1,2}_b (w,b interleaved)
 *   then per lvc block n (22 + 2*layers arrays): see fdo_lvc_block
 *   then final_w [1,C,7], final_b [1]
 * taps: NULL or an array of pointers (each may be NULL):
 *   0 embed [B,eout]; 1..nb skips a0..a_{nb-1}; nb+1 bottom a_nb; then per block: kernels, bias, x_out
 */
int fdo_num_weights(const fdo_config *cfg) { return 6 + cfg->n_blocks * 8 + cfg->n_blocks * (22 + 2 * cfg->lvc_layers) + 2; }

void fdo_forward(const fdo_config *cfg, const real *const *w, const float *embed_table,
                 const real *audio, const real *c, const real *steps, int B, int T, real *out, real *const *taps)
{
    const int C = cfg->inner_channels, nb = cfg->n_blocks;
    int hop_total = 1;
    for (int n = 0; n < nb; ++n) hop_total *= cfg->ratios[n];
    const int64_t L = (int64_t)T * hop_total;

    /* step embedding + 2-layer swish MLP (:85-87) */
    real *emb = (real *)malloc(sizeof(real) * (size_t)B * cfg->embed_in);
    real *e1 = (real *)malloc(sizeof(real) * (size_t)B * cfg->embed_mid);
    real *e2 = (real *)malloc(sizeof(real) * (size_t)B * cfg->embed_out);
    fdo_step_embedding(steps, B, embed_table, cfg->embed_in / 2, emb);
    fdo_linear(emb, B, cfg->embed_in, w[2], w[3], cfg->embed_mid, e1);
    fdo_swish(e1, (int64_t)B * cfg->embed_mid);
    fdo_linear(e1, B, cfg->embed_mid, w[4], w[5], cfg->embed_out, e2);
    fdo_swish(e2, (int64_t)B * cfg->embed_out);
    if (taps && taps[0]) memcpy(taps[0], e2, sizeof(real) * (size_t)B * cfg->embed_out);

    /* first_audio_conv (:89) and the down path (:90-93); downsample[d] has factor ratios[nb-1-d] (:63) */
    real *skips[FDO_MAX_BLOCKS + 1];
    int64_t lens[FDO_MAX_BLOCKS + 1];
    lens[0] = L;
    skips[0] = (real *)malloc(sizeof(real) * (size_t)B * C * L);
    fdo_conv1d(audio, B, 1, L, w[0], w[1], C, 7, 1, skips[0]);
    for (int d = 0; d < nb; ++d) {
        const int f = cfg->ratios[nb - 1 - d];
        lens[d + 1] = lens[d] / f;
        skips[d + 1] = (real *)malloc(sizeof(real) * (size_t)B * C * lens[d + 1]);
        fdo_dblock(skips[d], B, C, lens[d], f, w + 6 + 8 * d, skips[d + 1]);
    }
    if (taps)
        for (int d = 0; d <= nb; ++d)
            if (taps[1 + d]) memcpy(taps[1 + d], skips[d], sizeof(real) * (size_t)B * C * lens[d]);

    /* up path (:95-97): block n consumes skip reversed(downsample)[n] = skips[nb-1-n] */
    real *x = skips[nb];
    int64_t Lx = lens[nb];
    int hop = 1;
    const int per_blk = 22 + 2 * cfg->lvc_layers;
    for (int n = 0; n < nb; ++n) {
        hop *= cfg->ratios[n];
        const int64_t Lo = Lx * cfg->ratios[n];
        real *xo = (real *)malloc(sizeof(real) * (size_t)B * C * Lo);
        real *tk = taps ? taps[nb + 2 + 3 * n] : NULL, *tb = taps ? taps[nb + 3 + 3 * n] : NULL;
        fdo_lvc_block(cfg, n, x, skips[nb - 1 - n], c, e2, B, T, Lx, hop, w + 6 + 8 * nb + per_blk * n, xo, tk, tb);
        if (taps && taps[nb + 4 + 3 * n]) memcpy(taps[nb + 4 + 3 * n], xo, sizeof(real) * (size_t)B * C * Lo);
        if (n > 0) free(x);
        x = xo; Lx = Lo;
    }
    /* final_conv (:100) */
    const real *const *wf = w + 6 + 8 * nb + per_blk * nb;
    fdo_conv1d(x, B, C, L, wf[0], wf[1], 1, 7, 1, out);
    if (nb > 0) free(x);
    for (int d = 0; d <= nb; ++d) free(skips[d]);
    free(emb); free(e1); free(e2);
}

/* ------------------------------------------------------------------ */
/* a12/a13: schedule math -- fp32 exactly as the reference's torch ops */
/* ------------------------------------------------------------------ */

/* compute_hyperparams_given_schedule (util.py:365-390): sequential fp32 recursion, then sqrt */
void fdo_compute_hyperparams(const float *beta, int T, float *alpha, float *sigma)
{
    for (int t = 0; t < T; ++t) { alpha[t] = 1.0f - beta[t]; sigma[t] = beta[t] + 0.0f; }
    for (int t = 1; t < T; ++t) {
        volatile float a = alpha[t] * alpha[t - 1];
        alpha[t] = a;
        volatile float num = 1.0f - alpha[t - 1], den = 1.0f - alpha[t];
        volatile float q = num / den;
        volatile float s = sigma[t] * q;
        sigma[t] = s;
    }
    for (int t = 0; t < T; ++t) { alpha[t] = sqrtf(alpha[t]); sigma[t] = sqrtf(sigma[t]); }
}

/* map_noise_scale_to_time_step (util.py:394-404); returns -1 when no bracket is found */
double fdo_map_noise_scale_to_time_step(float alpha_infer, const float *alpha, int T)
{
    if (alpha_infer < alpha[T - 1]) return (double)(T - 1);
    if (alpha_infer > alpha[0]) return 0.0;
    for (int t = 0; t < T - 1; ++t)
        if (alpha[t + 1] <= alpha_infer && alpha_infer <= alpha[t]) {
            volatile float d = alpha[t] - alpha_infer;
            volatile float den = alpha[t] - alpha[t + 1];
            volatile float q = d / den;
            return (double)t + (double)q;
        }
    return -1.0;
}

/* The per-step scalars of sampling_given_noise_schedule (util.py:187-195,219-229), fp32 like the 0-d tensors there.
 * beta [N] -> alpha_hat [N], sigma_hat [N], c_eps [N] = beta/sqrt(1-alpha_hat^2), c_div [N] = sqrt(1-beta),
 * ddim c1, c2, c3 [N]. */
void fdo_inference_coefficients(const float *beta, int N, float *alpha_hat, float *sigma_hat, float *c_eps,
                                float *c_div, float *c1, float *c2, float *c3)
{
    fdo_compute_hyperparams(beta, N, alpha_hat, sigma_hat);   /* same recursion (:187-195) */
    for (int n = 0; n < N; ++n) {
        volatile float a2 = powf(alpha_hat[n], 2.0f);
        volatile float om = 1.0f - a2;
        volatile float sq = sqrtf(om);
        c_eps[n] = beta[n] / sq;                              /* :226 */
        volatile float omb = 1.0f - beta[n];
        c_div[n] = sqrtf(omb);                                /* :227 */
        volatile float an = alpha_hat[n] / c_div[n];          /* alpha_next (:220) */
        c1[n] = an / alpha_hat[n];                            /* :221 */
        volatile float nsq = -sq;
        c2[n] = nsq * c1[n];                                  /* :222 */
        volatile float an2 = powf(an, 2.0f);
        volatile float oman = 1.0f - an2;
        c3[n] = sqrtf(oman);                                  /* :223 */
    }
}

/* ------------------------------------------------------------------ */
/* a15: the reverse loop of sampling_given_noise_schedule (util.py:211-235) with injected noise */
/* ------------------------------------------------------------------ */
/* steps_t [N] mapped time steps; coefficient arrays [N] as above (passed as double, cast to real);
 * x_T [B,1,L] start; z [N,B,L] noise, z[n] used after step n when n>0 (z[0] unused);
 * seq (nullable) [N+1,B,L] trajectory as return_sequence=True gives it. */
void fdo_sample(const fdo_config *cfg, const real *const *w, const float *embed_table, const real *c,
                int B, int T, int N, const double *steps_t, const double *c_eps, const double *c_div,
                const double *sigma_hat, const double *c1, const double *c2, const double *c3, int ddim,
                const real *x_T, const real *z, real *out, real *seq)
{
    int hop_total = 1;
    for (int n = 0; n < cfg->n_blocks; ++n) hop_total *= cfg->ratios[n];
    const int64_t n_el = (int64_t)B * T * hop_total;
    real *x = out;
    real *eps = (real *)malloc(sizeof(real) * (size_t)n_el);
    real *st = (real *)malloc(sizeof(real) * (size_t)B);
    memcpy(x, x_T, sizeof(real) * (size_t)n_el);
    if (seq) memcpy(seq, x, sizeof(real) * (size_t)n_el);
    for (int n = N - 1, k = 1; n >= 0; --n, ++k) {
        for (int b = 0; b < B; ++b) st[b] = (real)steps_t[n];          /* :217 */
        fdo_forward(cfg, w, embed_table, x, c, st, B, T, eps, NULL);   /* :218 */
        if (ddim) {                                                    /* :219-224 */
            const real a = (real)c1[n], bb = (real)c2[n], cc = (real)c3[n];
            for (int64_t i = 0; i < n_el; ++i) {
                real t1 = a * x[i], t2 = bb * eps[i], t3 = cc * eps[i];
                x[i] = (t1 + t2) + t3;
            }
        } else {                                                       /* :226-229 */
            const real ce = (real)c_eps[n], cd = (real)c_div[n], sg = (real)sigma_hat[n];
            for (int64_t i = 0; i < n_el; ++i) {
                real v = x[i] - ce * eps[i];
                v = v / cd;
                if (n > 0) v = v + sg * z[(int64_t)n * n_el + i];
                x[i] = v;
            }
        }
        if (seq) memcpy(seq + (int64_t)k * n_el, x, sizeof(real) * (size_t)n_el);
    }
    free(eps); free(st);
}

/* ------------------------------------------------------------------ */
/* §8(f) next row 1: waveform epilogue (FastDiff.py:110-118, utils/audio.py:11-16) */
/* ------------------------------------------------------------------ */
/* wav_pred / wav_pred.abs().max() per utterance, then *32767 and astype(int16) (truncation toward zero) */
void fdo_peak_normalize_int16(const real *wav, int B, int64_t L, int16_t *pcm)
{
    for (int b = 0; b < B; ++b) {
        float m = 0.0f;
        for (int64_t i = 0; i < L; ++i) { float a = fabsf((float)wav[b * L + i]); if (a > m) m = a; }
        for (int64_t i = 0; i < L; ++i) {
            volatile float v = (float)wav[b * L + i] / m;
            volatile float s = v * 32767.0f;
            pcm[b * L + i] = (int16_t)s;
        }
    }
}
