// fd_kernels_cconv.hip -- the 21 small convolutions of the denoiser as ONE differentiable operator for the training path (SURVEY.md 8f
// row 4): the three `Conv1d(32, 32, 3, dilation = 1 / 2 / 4)` of every DiffusionDBlock (modules/FastDiff/module/modules.py:120-125,
// applied as `layer(leaky_relu(x, 0.2))`, :136-137) and the four `Conv1d(32, 32, 3, dilation = 3^i)` of every TimeAware_LVCBlock
// (modules.py:183-187, applied as `x += audio_down; y = leaky_relu(conv(leaky_relu(x, 0.2)), 0.2)`, :209-212):
//
//     xs = x (+ skip)                         (the layer's input; written out when there is a skip: the gate behind the LVC reads it)
//     y  = post(bias + W * pre(xs)),          pre(v) = leaky_relu(v, pre_slope),  post(v) = leaky_relu(v, post_slope) or the identity
//
// forward in one pass (read x, skip; write xs, y), and backward in one pass as well: from dy, y (the sign of the pre-activation), xs and
// the gradient that reached xs from its other readers it produces  dxs = gxs + mask_pre * (W^T * du),  du = dy * mask_post,  the weight
// gradient dW[o][i][k] = sum du[o][t] a[i][t + (k - 1) d]  and the bias gradient db[o] = sum du[o][t]  -- the two sums as per-workgroup
// partials (every workgroup walks its tiles in a fixed order and keeps the 32 x 96 tile of dW in matrix accumulators), added up in a
// fixed order by a second small kernel: same bits every run.  Under torch autograd the same layer is an add, two leaky-relus, a
// MIOpen convolution with NHWC transposes around it, and in the backward two more convolutions, two leaky-relu gradients, an add and
// a reduction launch for the bias: profiles/r03_train_step_families.txt.
// All products on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32): training keeps fp32 products, nothing to range-check.
// Weight layout as the reference's parameter: [out 32][in 32][tap 3] (already folded: g * v / ||v|| is the caller's, torch._weight_norm).
#include <algorithm>

#include "fd_kernels.h"

namespace fdk_cconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// D layout of the 32x32 tile: register r of lane (col = lane & 31, hi = lane >> 5) is row (r & 3) + 8 (r >> 2) + 4 hi
__device__ __forceinline__ int drow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.0f ? v : v * s; }
__device__ __forceinline__ float dlrelu(float v, float s) { return v > 0.0f ? 1.0f : s; }      // torch: grad * (x > 0 ? 1 : slope)

constexpr int C = 32, KS = 3, NW = C * C * KS;      // 3072 weights
constexpr int PSTRIDE = NW + C;                     // one workgroup's partial: dW [32][32][3] then db [32]

template <int DIL> struct Cfg {
    static constexpr int H = (DIL + 3) & ~3;                    // halo columns per side (a multiple of 4: float4 staging)
    static constexpr int WF = 256;                              // forward tile
    static constexpr int WB = 128;                              // backward tile (256 columns: the staging registers next to the two
                                                                // resident operand sets spill)
};

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: workgroup = 256 columns of one utterance, wave = 64 of them (two 32 x 32 tiles), K = 96 = (tap, in) in 48 steps
// ---------------------------------------------------------------------------------------------------------------------------------
template <int DIL>
__global__ void __launch_bounds__(256) k_cconv_fwd(const float *__restrict__ x, const float *__restrict__ skip, const float *__restrict__ w,
                                                   const float *__restrict__ bias, float *__restrict__ xs_out, float *__restrict__ y, int L,
                                                   int tiles_per_row, float pre, float post)
{
    constexpr int W = Cfg<DIL>::WF, H = Cfg<DIL>::H, XLD = W + 2 * H, NF4 = XLD / 4, TOTAL = C * NF4, NK = (TOTAL + 255) / 256;
    __shared__ __attribute__((aligned(16))) float as[C * XLD];      // pre(x + skip), column c at index c + H
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / tiles_per_row, w0 = (blockIdx.x - b * tiles_per_row) * W;
    // A operand: lane (row = out channel l31, half hi) holds k = 2 s + hi = tap * 32 + in for s = 0 .. 47
    float wa[48];
#pragma unroll
    for (int s = 0; s < 48; ++s) {
        const int kk = 2 * s + hi;
        wa[s] = w[(l31 * C + (kk & 31)) * KS + (kk >> 5)];
    }
    float cb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) cb[r] = bias[drow(r, hi)];
    {
        const float *xr = x + (int64_t)b * C * L, *sr = skip ? skip + (int64_t)b * C * L : nullptr;
        float *xo = xs_out ? xs_out + (int64_t)b * C * L : nullptr;
        float4 xa[NK], sa[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4, g = w0 - H + 4 * c4;
            const bool ok = idx < TOTAL && g >= 0 && g < L;      // L is a multiple of 4: a quad is all in or all out
            xa[k] = ok ? *reinterpret_cast<const float4 *>(xr + (int64_t)ci * L + g) : make_float4(0.f, 0.f, 0.f, 0.f);
            sa[k] = (ok && sr) ? *reinterpret_cast<const float4 *>(sr + (int64_t)ci * L + g) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4, g = w0 - H + 4 * c4;
            if (idx < TOTAL) {
                const float4 r = make_float4(xa[k].x + sa[k].x, xa[k].y + sa[k].y, xa[k].z + sa[k].z, xa[k].w + sa[k].w);
                if (xo && c4 >= H / 4 && c4 < H / 4 + W / 4 && g < L) *reinterpret_cast<float4 *>(xo + (int64_t)ci * L + g) = r;
                *reinterpret_cast<float4 *>(as + ci * XLD + 4 * c4) = make_float4(lrelu(r.x, pre), lrelu(r.y, pre), lrelu(r.z, pre), lrelu(r.w, pre));
            }
        }
    }
    __syncthreads();
    const int cw = wave * 64;
    if (w0 + cw >= L) return;
    float *yo = y + (int64_t)b * C * L + w0;
    const unsigned Lu = (unsigned)L;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int col = cw + ct * 32 + l31;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = cb[r];
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            const int kk = 2 * s, tap = kk >> 5, in0 = kk & 31;      // this lane's k = kk + hi: same tap, channel in0 + hi
            acc = mfma32(wa[s], as[(in0 + hi) * XLD + H + col + (tap - 1) * DIL], acc);
        }
        if (w0 + col < L) {
#pragma unroll
            for (int r = 0; r < 16; ++r) yo[(unsigned)drow(r, hi) * Lu + (unsigned)col] = lrelu(acc[r], post);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward: persistent workgroups over tiles (utterance, WB columns).  Per tile: du and a = pre(xs) with halo in LDS (odd row stride:
// the dW products read them with lane = channel); dxs for the tile's columns as 32 (in) x 96 (tap, out) x columns; this wave's share of
// dW as 32 (out) x 32 (in) per tap with k = its columns, kept in accumulators over all tiles of the workgroup.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int DIL>
__global__ void __launch_bounds__(256, 2) k_cconv_bwd(const float *__restrict__ xs, const float *__restrict__ y, const float *__restrict__ w,
                                                      const float *__restrict__ dy, const float *__restrict__ gxs, float *__restrict__ dxs,
                                                      float *__restrict__ partial, int L, int tiles_per_row, int ntiles, float pre, float post)
{
    constexpr int W = Cfg<DIL>::WB, H = Cfg<DIL>::H, XW = W + 2 * H, XLD = XW + 1 - (XW & 1) + 0, NF4 = XW / 4, TOTAL = C * NF4, NK = (TOTAL + 255) / 256;
    static_assert(XLD % 2 == 1, "odd row stride: lanes that index rows hit different banks");
    constexpr int WC = W / 4, NCT = WC / 32;           // columns per wave, 32-column tiles per wave
    constexpr int SMEM = (2 * C * XLD > 4 * NW + 4 * C) ? 2 * C * XLD : 4 * NW + 4 * C;      // the two images; at the end the reduction buffer
    __shared__ float smem[SMEM];
    float *du = smem;                                  // dy * post'(y), column c at index c + H, zero outside the signal
    float *aa = smem + C * XLD;                        // pre(xs), likewise
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    // A operand of the dxs product: lane (row = in channel l31, half hi) holds k' = 2 s + hi = tap * 32 + out: W[out][l31][tap]
    float wt[48];
#pragma unroll
    for (int s = 0; s < 48; ++s) {
        const int kk = 2 * s + hi;
        wt[s] = w[((kk & 31) * C + l31) * KS + (kk >> 5)];
    }
    f32x16 dw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[k][r] = 0.0f;
    float db = 0.0f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_row, w0 = (tile - b * tiles_per_row) * W;
        {
            const float *yr = y + (int64_t)b * C * L, *gr = dy + (int64_t)b * C * L, *xr = xs + (int64_t)b * C * L;
            // two batches (dy and y, then xs): all three at once would not fit the registers next to the two operand sets
            {
                float4 ya[NK], ga[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int idx = k * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4, g = w0 - H + 4 * c4;
                    const bool ok = idx < TOTAL && g >= 0 && g < L;
                    const int64_t off = (int64_t)ci * L + g;
                    ya[k] = ok ? *reinterpret_cast<const float4 *>(yr + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                    ga[k] = ok ? *reinterpret_cast<const float4 *>(gr + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int idx = k * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4;
                    if (idx < TOTAL) {
                        float *d = du + ci * XLD + 4 * c4;
                        d[0] = ga[k].x * dlrelu(ya[k].x, post); d[1] = ga[k].y * dlrelu(ya[k].y, post);
                        d[2] = ga[k].z * dlrelu(ya[k].z, post); d[3] = ga[k].w * dlrelu(ya[k].w, post);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the second batch's loads behind the first batch's LDS writes
            {
                float4 xa[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int idx = k * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4, g = w0 - H + 4 * c4;
                    const bool ok = idx < TOTAL && g >= 0 && g < L;
                    xa[k] = ok ? *reinterpret_cast<const float4 *>(xr + (int64_t)ci * L + g) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int idx = k * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4;
                    if (idx < TOTAL) {
                        float *a = aa + ci * XLD + 4 * c4;
                        a[0] = lrelu(xa[k].x, pre); a[1] = lrelu(xa[k].y, pre); a[2] = lrelu(xa[k].z, pre); a[3] = lrelu(xa[k].w, pre);
                    }
                }
            }
        }
        __syncthreads();
        const int cw = wave * WC;
        if (w0 + cw < L) {
            // ---- dxs[i][t] = gxs[i][t] + pre'(xs[i][t]) * sum_{tap, o} W[o][i][tap] du[o][t - (tap - 1) d]
            if (dxs) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int col = cw + ct * 32 + l31;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                    for (int s = 0; s < 48; ++s) {
                        const int kk = 2 * s, tap = kk >> 5, o0 = kk & 31;
                        acc = mfma32(wt[s], du[(o0 + hi) * XLD + H + col - (tap - 1) * DIL], acc);
                    }
                    if (w0 + col < L) {
                        // uniform 64-bit bases, 32-bit per-lane offsets (32 L < 2^31): sixteen rows of addresses would otherwise cost 64 registers
                        const float *gb = gxs ? gxs + (int64_t)b * C * L + w0 : nullptr;
                        float *ob = dxs + (int64_t)b * C * L + w0;
                        const unsigned Lu = (unsigned)L, o0 = (unsigned)(4 * hi) * Lu + (unsigned)col;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int i = drow(r, hi);
                            const unsigned off = o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * Lu;
                            // a > 0 <=> xs > 0 (leaky-relu keeps the sign; a = 0 <=> xs = 0, where torch's gradient is the slope)
                            const float m = aa[i * XLD + H + col] > 0.0f ? 1.0f : pre;
                            const float g0 = gb ? gb[off] : 0.0f;
                            ob[off] = fmaf(acc[r], m, g0);
                        }
                    }
                }
            }
            // ---- dW[o][i][tap] += sum_t du[o][t] a[i][t + (tap - 1) d] over this wave's columns; db[o] += sum_t du[o][t]
#pragma unroll 4
            for (int s = 0; s < WC / 2; ++s) {
                const int t = H + cw + 2 * s + hi;
                const float av = du[l31 * XLD + t];
                db += av;
#pragma unroll
                for (int k = 0; k < 3; ++k) dw[k] = mfma32(av, aa[l31 * XLD + t + (k - 1) * DIL], dw[k]);
            }
        }
        __syncthreads();
    }
    // ---- the workgroup's partial: the four waves' accumulators added in a fixed order through LDS ---------------------------------
    float *red = smem;                                  // [wave][tap][out][in] = 4 x 3072 floats over the (dead) images
    float *redb = smem + 4 * NW;                        // [wave][32]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 3 + k) * C + drow(r, hi)) * C + l31] = dw[k][r];
    db += __shfl_xor(db, 32, 64);
    if (hi == 0) redb[wave * C + l31] = db;
    __syncthreads();
    float *pout = partial + (int64_t)blockIdx.x * PSTRIDE;
    for (int e = tid; e < NW; e += 256) {              // e = (tap * 32 + out) * 32 + in  ->  weight layout [out][in][tap]
        const int k = e / (C * C), o = (e / C) % C, i = e % C;
        pout[(o * C + i) * KS + k] = (red[e] + red[NW + e]) + (red[2 * NW + e] + red[3 * NW + e]);
    }
    if (tid < C) pout[NW + tid] = (redb[tid] + redb[C + tid]) + (redb[2 * C + tid] + redb[3 * C + tid]);
}

// partial [nparts][PSTRIDE] -> dW [3072], db [32]: workgroup = 32 elements x 8 slices of the partials, each slice summed in order, the
// eight slice sums added in order
// (partials pstride floats apart, `count` of them meaningful: the first nw are the weight gradient, the rest the bias gradient)
__global__ void __launch_bounds__(256) k_cconv_reduce(const float *__restrict__ partial, int nparts, int pstride, int count, int nw,
                                                      float *__restrict__ dw, float *__restrict__ db)
{
    __shared__ float sl[8][32];
    const int e = blockIdx.x * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
    const int per = (nparts + 7) / 8, p0 = q * per, p1 = min(nparts, p0 + per);
    float acc = 0.0f;
    if (e < count) {      // four running sums (four loads in flight instead of one), joined in a fixed order
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int p = p0;
        for (; p + 3 < p1; p += 4) {
            a0 += partial[(int64_t)p * pstride + e];
            a1 += partial[(int64_t)(p + 1) * pstride + e];
            a2 += partial[(int64_t)(p + 2) * pstride + e];
            a3 += partial[(int64_t)(p + 3) * pstride + e];
        }
        for (; p < p1; ++p) a0 += partial[(int64_t)p * pstride + e];
        acc = (a0 + a1) + (a2 + a3);
    }
    sl[q][threadIdx.x & 31] = acc;
    __syncthreads();
    if (q == 0 && e < count) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += sl[j][threadIdx.x];
        if (e < nw) { if (dw) dw[e] = s; }
        else if (db) db[e - nw] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The two 7-tap convolutions at the ends of the network on the training path: first_audio_conv = Conv1d(1, 32, 7, pad 3)
// (FastDiff_model.py:34-36,89) and final_conv = Conv1d(32, 1, 7, pad 3) (FastDiff_model.py:67-68,100).  K = 7 is nothing for the matrix
// pipe: plain VALU, tiles of 256 columns through LDS, the weights through vector loads + LDS (never scalar loads: LABBOOK.md section 4).
// Both weight gradients are the same correlation of a 32-row tile R with ONE row s,  C[r][k] = sum_t R[r][t] s[t + sgn (k - 3)]:
//   first conv:  R = dy [32 rows], s = x,  sgn = +1;      final conv:  R = x [32 rows], s = dy,  sgn = -1
// thread = (row r, eighth of the tile's columns, interleaved), 7 running sums kept over all tiles of the persistent workgroup, the
// eighths added by a butterfly, per-workgroup partials added in a fixed order by k_cconv_reduce.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int C7W = 256, C7LD = 264;      // tile width; LDS row stride (264 = 8 mod 32: lanes (row 0..7, eighth 0..7) hit 64 different banks)
constexpr int C7P = 32 * 7 + 32;          // one workgroup's partial: dW [32][7], then up to 32 bias sums

__global__ void __launch_bounds__(256) k_c7_first_fwd(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                      float *__restrict__ y, int L)
{
    __shared__ float wl[256];      // [out][7 taps + bias]
    wl[threadIdx.x] = (threadIdx.x & 7) < 7 ? w[(threadIdx.x >> 3) * 7 + (threadIdx.x & 7)] : bias[threadIdx.x >> 3];
    __syncthreads();
    const int b = blockIdx.y, t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t0 >= L) return;
    const float *xr = x + (int64_t)b * L;
    float xv[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int p = t0 - 3 + i;
        xv[i] = (p >= 0 && p < L) ? xr[p] : 0.0f;
    }
#pragma unroll 4
    for (int o = 0; o < C; ++o) {
        const float bv = wl[o * 8 + 7];
        float4 r = make_float4(bv, bv, bv, bv);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float wv = wl[o * 8 + k];
            r.x += wv * xv[k]; r.y += wv * xv[k + 1]; r.z += wv * xv[k + 2]; r.w += wv * xv[k + 3];
        }
        *reinterpret_cast<float4 *>(y + ((int64_t)b * C + o) * L + t0) = r;
    }
}

// stage rows [32][C7W + 6] of R (columns w0 - 3 .. w0 + C7W + 2, zero outside [0, L)) and the same window of s
__device__ __forceinline__ void c7_stage(float *__restrict__ Rs, float *__restrict__ ss, const float *__restrict__ R, const float *__restrict__ s1,
                                         int b, int w0, int L, int tid)
{
    for (int idx = tid; idx < C * (C7W + 6); idx += 256) {
        const int r = idx / (C7W + 6), c = idx - r * (C7W + 6), g = w0 - 3 + c;
        Rs[r * C7LD + c] = (g >= 0 && g < L) ? R[((int64_t)b * C + r) * L + g] : 0.0f;
    }
    for (int c = tid; c < C7W + 6; c += 256) {
        const int g = w0 - 3 + c;
        ss[c] = (g >= 0 && g < L) ? s1[(int64_t)b * L + g] : 0.0f;
    }
}

// MODE 0 = first conv backward (R = dy, s = x): dW, db[32] = row sums of dy, and (optionally) dx[t] = sum_{o,k} W[o][k] dy[o][t + 3 - k]
// MODE 1 = final conv backward (R = x, s = dy): dW, db[1] = sum of dy, and dx[i][t] = sum_k W[i][k] dy[t + 3 - k]
template <int MODE>
__global__ void __launch_bounds__(256) k_c7_bwd(const float *__restrict__ R, const float *__restrict__ s1, const float *__restrict__ w,
                                                float *__restrict__ dx, float *__restrict__ partial, int L, int tiles_per_row, int ntiles)
{
    __shared__ float Rs[C * C7LD];
    __shared__ float ss[C7W + 8];
    __shared__ float wl[C * 8];
    const int tid = threadIdx.x, r = tid >> 3, part = tid & 7;
    wl[tid] = part < 7 ? w[r * 7 + part] : 0.0f;      // weight of row r (first conv: out channel, final conv: in channel), tap `part`
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float rowsum = 0.0f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_row, w0 = (tile - b * tiles_per_row) * C7W;
        __syncthreads();
        c7_stage(Rs, ss, R, s1, b, w0, L, tid);
        __syncthreads();
        // ---- the correlation: this thread's columns part, part + 8, ... of the tile (centre columns: LDS index 3 + t)
#pragma unroll 4
        for (int j = 0; j < C7W / 8; ++j) {
            const int t = part + 8 * j;
            const float rv = Rs[r * C7LD + 3 + t];
            if (MODE == 0) rowsum += rv;
            else if (r == 0) rowsum += ss[3 + t];
#pragma unroll
            for (int k = 0; k < 7; ++k) acc[k] = fmaf(rv, ss[3 + t + (MODE == 0 ? k - 3 : 3 - k)], acc[k]);
        }
        // ---- the input gradient
        if (dx) {
            const int t = tid, g = w0 + t;
            if (MODE == 0) {
                if (g < L) {
                    float v = 0.0f;
#pragma unroll 4
                    for (int o = 0; o < C; ++o)
#pragma unroll
                        for (int k = 0; k < 7; ++k) v = fmaf(wl[o * 8 + k], Rs[o * C7LD + 3 + t + 3 - k], v);
                    dx[(int64_t)b * L + g] = v;
                }
            } else {
                if (g < L) {
                    float dv[7];
#pragma unroll
                    for (int k = 0; k < 7; ++k) dv[k] = ss[3 + t + 3 - k];
#pragma unroll 4
                    for (int i = 0; i < C; ++i) {
                        float v = 0.0f;
#pragma unroll
                        for (int k = 0; k < 7; ++k) v = fmaf(wl[i * 8 + k], dv[k], v);
                        dx[((int64_t)b * C + i) * L + g] = v;
                    }
                }
            }
        }
    }
    // the eighths of a row sit in eight neighbouring lanes: butterfly, then lane part == 0 writes the workgroup's partial
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        acc[k] += __shfl_xor(acc[k], 1, 64);
        acc[k] += __shfl_xor(acc[k], 2, 64);
        acc[k] += __shfl_xor(acc[k], 4, 64);
    }
    rowsum += __shfl_xor(rowsum, 1, 64);
    rowsum += __shfl_xor(rowsum, 2, 64);
    rowsum += __shfl_xor(rowsum, 4, 64);
    float *pout = partial + (int64_t)blockIdx.x * C7P;
    if (part == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) pout[r * 7 + k] = acc[k];
        if (MODE == 0 || r == 0) pout[C * 7 + r] = rowsum;
    }
}

__global__ void __launch_bounds__(256) k_c7_final_fwd(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                      float *__restrict__ y, int L, int tiles_per_row)
{
    __shared__ float Rs[C * C7LD];
    __shared__ float wl[C * 8];
    const int tid = threadIdx.x, b = blockIdx.x / tiles_per_row, w0 = (blockIdx.x - b * tiles_per_row) * C7W;
    wl[tid] = (tid & 7) < 7 ? w[(tid >> 3) * 7 + (tid & 7)] : (tid == 7 ? bias[0] : 0.0f);      // the bias rides in slot 7 of row 0
    for (int idx = tid; idx < C * (C7W + 6); idx += 256) {
        const int r = idx / (C7W + 6), c = idx - r * (C7W + 6), g = w0 - 3 + c;
        Rs[r * C7LD + c] = (g >= 0 && g < L) ? x[((int64_t)b * C + r) * L + g] : 0.0f;
    }
    __syncthreads();
    const int g = w0 + tid;
    if (g >= L) return;
    float v = wl[7];
#pragma unroll 4
    for (int i = 0; i < C; ++i)
#pragma unroll
        for (int k = 0; k < 7; ++k) v = fmaf(wl[i * 8 + k], Rs[i * C7LD + tid + k], v);
    y[(int64_t)b * L + g] = v;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// weight-norm (FastDiff_model.py:115-122: torch.nn.utils.weight_norm on every Conv1d = torch._weight_norm(v, g, 0)):
//   w[r, :] = v[r, :] * g[r] / ||v[r, :]||,   and from dw:   dg[r] = <dw[r], v[r]> / ||v[r]||,   dv[r] = (g / ||v||) (dw[r] - v[r] <dw[r], v[r]> / ||v||^2)
// one wave per row (row = output channel, cols = in * k: 7 .. 400), sums by a butterfly: the same order every run
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__global__ void __launch_bounds__(256) k_wn_fwd(const float *__restrict__ v, const float *__restrict__ g, float *__restrict__ w,
                                                float *__restrict__ norm, int64_t rows, int cols)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *vr = v + r * cols;
    float ss = 0.0f;
    for (int c = lane; c < cols; c += 64) ss = fmaf(vr[c], vr[c], ss);
    const float nrm = sqrtf(wave_sum(ss)), sc = g[r] / nrm;
    for (int c = lane; c < cols; c += 64) w[r * cols + c] = vr[c] * sc;
    if (lane == 0) norm[r] = nrm;
}

__global__ void __launch_bounds__(256) k_wn_bwd(const float *__restrict__ v, const float *__restrict__ g, const float *__restrict__ norm,
                                                const float *__restrict__ dw, float *__restrict__ dv, float *__restrict__ dg, int64_t rows, int cols)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *vr = v + r * cols, *dr = dw + r * cols;
    float dot = 0.0f;
    for (int c = lane; c < cols; c += 64) dot = fmaf(dr[c], vr[c], dot);
    dot = wave_sum(dot);
    const float nrm = norm[r], gn = g[r] / nrm, k2 = dot / (nrm * nrm);
    for (int c = lane; c < cols; c += 64) dv[r * cols + c] = gn * (dr[c] - vr[c] * k2);
    if (lane == 0) dg[r] = dot / nrm;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The block's up-sampler on the training path: `self.upsample(F.leaky_relu(x, 0.2))` with upsample = ConvTranspose1d(32, 32, 2r, stride r,
// padding r / 2) (modules.py:163-166,205-206), r = 8, 8, 4:
//     a = leaky_relu(x, 0.2);   y[o][c] = bias[o] + sum_i sum_j a[i][j] W[i][o][c + r/2 - j r]        (taps 0 <= c + r/2 - j r < 2 r)
// Output phase ph = c mod r, position q = c / r: two input positions contribute, q + offA with tap kA and q + offA - 1 with tap kA + r,
//     ph <  r/2:  offA = 0, kA = ph + r/2;        ph >= r/2:  offA = 1, kA = ph - r/2
// Backward (one kernel): tap k reaches column c = j r + k - r/2 = (j + qoff(k)) r + ph(k),  ph(k) = (k - r/2) mod r,  qoff(k) = floor((k - r/2) / r):
//     da[i][j] = sum_{k, o} W[i][o][k] dy[o][(j + qoff(k)) r + ph(k)],   dx = da * leaky_relu'(x),
//     dW[i][o][k] = sum_{b, j} a[i][j] dy[o][(j + qoff(k)) r + ph(k)],   db[o] = sum dy[o][.]
// Tile = 256 output columns = Q = 256 / r input positions.  LDS: dy phase-major [o][phase][position -1 .. Q] (a matrix tile's 32 lanes
// then read 32 neighbouring floats whichever operand they feed), a [i][position -1 .. Q] (odd strides where lanes index rows).
// Wave w owns the taps k in [w r/2, (w + 1) r/2): its share of dW (r/2 accumulator tiles of 32 x 32, kept over all tiles of the
// persistent workgroup) and its share of the da sum, which the four waves add up through LDS in a fixed order.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int R> struct CtCfg {
    static constexpr int Q = 256 / R, NT = Q / 32, KW = R / 2;      // positions per tile, 32-position tiles, taps per wave
    static constexpr int ALD = Q + 3;                                // a row: positions -1 .. Q, odd stride
    static constexpr int PLD = Q + 2, DLD = R * PLD + 1;             // dy: [phase][position -1 .. Q] per channel, odd channel stride
    static constexpr int PW = 2 * R * C * C;                         // weights
};

template <int R>
__global__ void __launch_bounds__(256) k_ct_fwd(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                float *__restrict__ y, int Lin, int tiles_per_row)
{
    using G = CtCfg<R>;
    constexpr int Q = G::Q, NT = G::NT, ALD = G::ALD, OLD = 256 + 4;
    __shared__ float as[C * ALD];                                     // leaky_relu(x), position j at index j - q0 + 1
    __shared__ __attribute__((aligned(16))) float os[C * OLD];        // the output tile, written back as whole rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / tiles_per_row, q0 = (blockIdx.x - b * tiles_per_row) * Q, Lout = Lin * R;
    for (int idx = tid; idx < C * (Q + 2); idx += 256) {
        const int i = idx / (Q + 2), jj = idx - i * (Q + 2), j = q0 - 1 + jj;
        as[i * ALD + jj] = (j >= 0 && j < Lin) ? lrelu(x[((int64_t)b * C + i) * Lin + j], 0.2f) : 0.0f;
    }
    __syncthreads();
    // units (phase, 32-position tile): R * NT = 8 of them, two per wave
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2) {
        const int u = wave * 2 + u2, ph = u / NT, pt = u - ph * NT;
        const int offA = ph < R / 2 ? 0 : 1, kA = ph < R / 2 ? ph + R / 2 : ph - R / 2;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias[drow(r, hi)];
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {      // k' = 2 s + hi = sel * 32 + i: sel 0 -> position q + offA, tap kA; sel 1 -> q + offA - 1, tap kA + r
            const int kk = 2 * s + hi, sel = kk >> 5, i = kk & 31;
            const float av = w[((int64_t)i * C + l31) * (2 * R) + kA + sel * R];
            acc = mfma32(av, as[i * ALD + 1 + pt * 32 + l31 + offA - sel], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) os[drow(r, hi) * OLD + (pt * 32 + l31) * R + ph] = acc[r];
    }
    __syncthreads();
    const int c0 = q0 * R;
    for (int idx = tid; idx < C * 64; idx += 256) {
        const int o = idx >> 6, c4 = (idx & 63) * 4;
        if (c0 + c4 < Lout) *reinterpret_cast<float4 *>(y + ((int64_t)b * C + o) * Lout + c0 + c4) = *reinterpret_cast<const float4 *>(os + o * OLD + c4);
    }
}

template <int R>
__global__ void __launch_bounds__(256) k_ct_bwd(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ dy,
                                                float *__restrict__ dx, float *__restrict__ partial, int Lin, int tiles_per_row, int ntiles)
{
    using G = CtCfg<R>;
    constexpr int Q = G::Q, NT = G::NT, KW = G::KW, ALD = G::ALD, PLD = G::PLD, DLD = G::DLD, PW = G::PW, P = R / 2;
    extern __shared__ float smem[];
    float *ws = smem;                       // [k][o][i]: the A operand of the da product, lane = i
    float *as = ws + PW;                    // [i][ALD]
    float *ds = as + C * ALD;               // [o][DLD]; at the end of a tile the four waves' da partials: [wave][NT][32 i][32 j]
    static_assert(4 * NT * 1024 <= C * DLD, "reduction buffer inside the dy image");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5, Lout = Lin * R;
    for (int idx = tid; idx < PW; idx += 256) {      // w [i][o][k] -> ws [k][o][i]
        const int k = idx / (C * C), o = (idx / C) % C, i = idx % C;
        ws[idx] = w[((int64_t)i * C + o) * (2 * R) + k];
    }
    f32x16 dwacc[KW];
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dwacc[t][r] = 0.0f;
    float dbacc = 0.0f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_row, q0 = (tile - b * tiles_per_row) * Q;
        __syncthreads();
        for (int idx = tid; idx < C * (Q + 2); idx += 256) {
            const int i = idx / (Q + 2), jj = idx - i * (Q + 2), j = q0 - 1 + jj;
            as[i * ALD + jj] = (j >= 0 && j < Lin) ? x[((int64_t)b * C + i) * Lin + j] : 0.0f;      // RAW x: the mask needs its sign; activated on use
        }
        for (int idx = tid; idx < C * R * PLD; idx += 256) {      // columns (q0 - 1) r .. (q0 + Q + 1) r - 1, consecutive threads = consecutive columns
            const int o = idx / (R * PLD), cc = idx - o * (R * PLD), qq = cc / R, ph = cc - qq * R, c = (q0 - 1 + qq) * R + ph;
            ds[o * DLD + ph * PLD + qq] = (c >= 0 && c < Lout) ? dy[((int64_t)b * C + o) * Lout + c] : 0.0f;
        }
        __syncthreads();
        // ---- this wave's taps: dW[.][.][k] += a^T-product over the tile's positions; db from the taps with qoff = 0 (each phase once)
        f32x16 da[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) da[t][r] = 0.0f;
#pragma unroll
        for (int t = 0; t < KW; ++t) {
            const int k = wave * KW + t, m = k - P, qoff = (m >= 0) ? m / R : -1, ph = m - qoff * R;
            const float *dp = ds + ph * PLD + 1 + qoff;                  // + o * DLD + j
            // dW tile: rows i (A = a[i][j], lane = i), cols o (B = dy[o][..], lane = o), k-steps = positions j
#pragma unroll 4
            for (int s = 0; s < Q / 2; ++s) {
                const int j = 2 * s + hi;
                const float bv = dp[l31 * DLD + j];
                if (qoff == 0) dbacc += bv;
                dwacc[t] = mfma32(lrelu(as[l31 * ALD + 1 + j], 0.2f), bv, dwacc[t]);
            }
            // da partial: rows i (A = W[i][o][k] = ws[k][o][i]), cols j (B = dy[o][..][j + qoff]), k-steps = o
#pragma unroll
            for (int pt = 0; pt < NT; ++pt)
#pragma unroll 4
                for (int s = 0; s < 16; ++s) {
                    const int o = 2 * s + hi;
                    da[pt] = mfma32(ws[(k * C + o) * C + l31], dp[o * DLD + pt * 32 + l31], da[pt]);
                }
        }
        __syncthreads();                                                  // every wave is done with the dy image
        if (dx) {
#pragma unroll
            for (int pt = 0; pt < NT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[((wave * NT + pt) * 32 + drow(r, hi)) * 32 + l31] = da[pt][r];
            __syncthreads();
            for (int idx = tid; idx < C * Q; idx += 256) {
                const int i = idx / Q, jl = idx - i * Q, j = q0 + jl;
                if (j < Lin) {
                    const int e = ((jl >> 5) * 32 + i) * 32 + (jl & 31);
                    const float v = (ds[e] + ds[NT * 1024 + e]) + (ds[2 * NT * 1024 + e] + ds[3 * NT * 1024 + e]);
                    dx[((int64_t)b * C + i) * Lin + j] = v * dlrelu(as[i * ALD + 1 + jl], 0.2f);
                }
            }
        }
    }
    __syncthreads();
    // ---- the workgroup's partial: dW [i][o][k] (the torch layout of a ConvTranspose1d weight), then db [o]
    float *pout = partial + (int64_t)blockIdx.x * (PW + C);
#pragma unroll
    for (int t = 0; t < KW; ++t) {
        const int k = wave * KW + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) pout[((int64_t)drow(r, hi) * C + l31) * (2 * R) + k] = dwacc[t][r];
    }
    dbacc += __shfl_xor(dbacc, 32, 64);
    float *red = smem;                                                    // (the weights are dead)
    if (hi == 0) red[wave * C + l31] = dbacc;
    __syncthreads();
    if (tid < C) pout[PW + tid] = (red[tid] + red[C + tid]) + (red[2 * C + tid] + red[3 * C + tid]);
}

}  // namespace fdk_cconv

namespace fdk {

using namespace fdk_cconv;

// ---- the same for MANY parameter tensors in one launch (the module has 53 weight-normed convolutions: 106 launches of a few
//      microseconds each per training step otherwise).  The records travel as KERNEL ARGUMENTS (<= WN_CHUNK per launch: 2.1 KB of the
//      4 KB a launch may carry), not through a table in device memory: nothing to upload, nothing whose lifetime a captured graph
//      would depend on.  first_block: the first workgroup of each tensor (4 rows per workgroup); a workgroup finds its tensor by
//      bisection.
constexpr int WN_CHUNK = 28;
struct WnChunk {
    fd_wn_item it[WN_CHUNK];
    int first_block[WN_CHUNK + 1];
    int n;
};

__device__ __forceinline__ int wn_find_item(const WnChunk &c, int blk)
{
    int lo = 0, hi = c.n;      // first_block[lo] <= blk < first_block[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (c.first_block[mid] <= blk) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256) k_wn_multi_fwd(const WnChunk c)
{
    const int it = wn_find_item(c, blockIdx.x);
    const fd_wn_item &I = c.it[it];
    const int64_t r = (int64_t)(blockIdx.x - c.first_block[it]) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, cols = I.cols;
    if (r >= I.rows) return;
    const float *vr = I.v + r * cols;
    float ss = 0.0f;
    for (int k = lane; k < cols; k += 64) ss = fmaf(vr[k], vr[k], ss);
    const float nrm = sqrtf(wave_sum(ss)), sc = I.g[r] / nrm;
    for (int k = lane; k < cols; k += 64) I.w[r * cols + k] = vr[k] * sc;
    if (lane == 0) I.norm[r] = nrm;
}

__global__ void __launch_bounds__(256) k_wn_multi_bwd(const WnChunk c)
{
    const int it = wn_find_item(c, blockIdx.x);
    const fd_wn_item &I = c.it[it];
    const int64_t r = (int64_t)(blockIdx.x - c.first_block[it]) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, cols = I.cols;
    if (r >= I.rows) return;
    if (!I.dw) {      // this weight took no part in the loss: zero gradients
        for (int k = lane; k < cols; k += 64) I.dv[r * cols + k] = 0.0f;
        if (lane == 0) I.dg[r] = 0.0f;
        return;
    }
    const float *vr = I.v + r * cols, *dr = I.dw + r * cols;
    float dot = 0.0f;
    for (int k = lane; k < cols; k += 64) dot = fmaf(dr[k], vr[k], dot);
    dot = wave_sum(dot);
    const float nrm = I.norm[r], gn = I.g[r] / nrm, k2 = dot / (nrm * nrm);
    for (int k = lane; k < cols; k += 64) I.dv[r * cols + k] = gn * (dr[k] - vr[k] * k2);
    if (lane == 0) I.dg[r] = dot / nrm;
}

// items: HOST memory
hipError_t weight_norm_multi(const Launch &L_, const fd_wn_item *items, int n, bool backward)
{
    for (int i0 = 0; i0 < n; i0 += WN_CHUNK) {
        WnChunk c;
        c.n = std::min(WN_CHUNK, n - i0);
        int blocks = 0;
        for (int i = 0; i < c.n; ++i) {
            c.it[i] = items[i0 + i];
            c.first_block[i] = blocks;
            blocks += (int)((items[i0 + i].rows + 3) / 4);
        }
        for (int i = c.n; i <= WN_CHUNK; ++i) c.first_block[i] = blocks;
        if (blocks == 0) continue;
        if (backward) FD_LAUNCH(L_, "weight_norm_multi_bwd", k_wn_multi_bwd, dim3((unsigned)blocks), dim3(256), 0, c);
        else FD_LAUNCH(L_, "weight_norm_multi_fwd", k_wn_multi_fwd, dim3((unsigned)blocks), dim3(256), 0, c);
    }
    return hipSuccess;
}

hipError_t weight_norm_forward(const Launch &L_, const float *v, const float *g, float *w, float *norm, int64_t rows, int cols)
{
    FD_LAUNCH(L_, "weight_norm_fwd", k_wn_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, v, g, w, norm, rows, cols);
    return hipSuccess;
}
hipError_t weight_norm_backward(const Launch &L_, const float *v, const float *g, const float *norm, const float *dw, float *dv, float *dg,
                                int64_t rows, int cols)
{
    FD_LAUNCH(L_, "weight_norm_bwd", k_wn_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, v, g, norm, dw, dv, dg, rows, cols);
    return hipSuccess;
}

// the block's up-sampler: leaky_relu(x, 0.2) -> ConvTranspose1d(32, 32, 2 r, stride r, padding r / 2), r = 4 or 8
static int ct_grid(const Launch &L_, int r, int B, int64_t Lin)
{
    const int Q = 256 / r;
    return (int)std::min<int64_t>((int64_t)B * ((Lin + Q - 1) / Q), (int64_t)L_.ctx->num_cus);
}
size_t convt_scratch_floats(const Launch &L_, int r, int B, int64_t Lin) { return (size_t)ct_grid(L_, r, B, Lin) * (2 * r * C * C + C); }

hipError_t convt_forward(const Launch &L_, const float *x, const float *w, const float *bias, float *y, int B, int64_t Lin, int r)
{
    const int Q = 256 / r, tiles = (int)((Lin + Q - 1) / Q);
    if (r == 4) FD_LAUNCH(L_, "convt_train_fwd", k_ct_fwd<4>, dim3(B * tiles), dim3(256), 0, x, w, bias, y, (int)Lin, tiles);
    else if (r == 8) FD_LAUNCH(L_, "convt_train_fwd", k_ct_fwd<8>, dim3(B * tiles), dim3(256), 0, x, w, bias, y, (int)Lin, tiles);
    else return hipErrorInvalidValue;
    return hipSuccess;
}

hipError_t convt_backward(const Launch &L_, const float *x, const float *w, const float *dy, float *dx, float *dw, float *db, int B, int64_t Lin,
                          int r, float *scratch)
{
    const int Q = 256 / r, tiles = (int)((Lin + Q - 1) / Q), ntiles = B * tiles, grid = ct_grid(L_, r, B, Lin);
    const int pw = 2 * r * C * C;
    if (r == 4) {
        using G = CtCfg<4>;
        const size_t shmem = sizeof(float) * (G::PW + C * G::ALD + C * G::DLD);
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ct_bwd<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (ea != hipSuccess) return ea;
        FD_LAUNCH(L_, "convt_train_bwd", k_ct_bwd<4>, dim3(grid), dim3(256), shmem, x, w, dy, dx, scratch, (int)Lin, tiles, ntiles);
    } else if (r == 8) {
        using G = CtCfg<8>;
        const size_t shmem = sizeof(float) * (G::PW + C * G::ALD + C * G::DLD);
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ct_bwd<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (ea != hipSuccess) return ea;
        FD_LAUNCH(L_, "convt_train_bwd", k_ct_bwd<8>, dim3(grid), dim3(256), shmem, x, w, dy, dx, scratch, (int)Lin, tiles, ntiles);
    } else return hipErrorInvalidValue;
    if (dw || db) FD_LAUNCH(L_, "cconv_reduce", k_cconv_reduce, dim3((pw + C + 31) / 32), dim3(256), 0, (const float *)scratch, grid, pw + C, pw + C, pw, dw, db);
    return hipSuccess;
}

bool cconv_supported(int dil, int64_t L) { return (dil == 1 || dil == 2 || dil == 3 || dil == 4 || dil == 9 || dil == 27) && L >= 4 && L % 4 == 0 && L < ((int64_t)1 << 25); }

int cconv_bwd_grid(const Launch &L_, int dil, int B, int64_t L)
{
    const int W = 128;
    const int64_t ntiles = (int64_t)B * ((L + W - 1) / W);
    return (int)std::min<int64_t>(ntiles, 2 * (int64_t)L_.ctx->num_cus);
}
size_t cconv_scratch_floats(const Launch &L_, int dil, int B, int64_t L) { return (size_t)cconv_bwd_grid(L_, dil, B, L) * PSTRIDE; }

hipError_t cconv_forward(const Launch &L_, const float *x, const float *skip, const float *w, const float *bias, float *xs_out, float *y, int B,
                         int64_t L, int dil, float pre, float post)
{
    const int tiles = (int)((L + 255) / 256);
#define FD_CC_FWD(D)                                                                                                                  \
    case D: FD_LAUNCH(L_, "cconv_fwd", k_cconv_fwd<D>, dim3(B * tiles), dim3(256), 0, x, skip, w, bias, xs_out, y, (int)L, tiles, pre, post); break
    switch (dil) {
        FD_CC_FWD(1); FD_CC_FWD(2); FD_CC_FWD(3); FD_CC_FWD(4); FD_CC_FWD(9); FD_CC_FWD(27);
    default: return hipErrorInvalidValue;
    }
#undef FD_CC_FWD
    return hipSuccess;
}

hipError_t cconv_backward(const Launch &L_, const float *xs, const float *y, const float *w, const float *dy, const float *gxs, float *dxs,
                          float *dw, float *db, int B, int64_t L, int dil, float pre, float post, float *scratch)
{
    const int W = 128, tiles = (int)((L + W - 1) / W), ntiles = B * tiles, grid = cconv_bwd_grid(L_, dil, B, L);
#define FD_CC_BWD(D)                                                                                                                  \
    case D: FD_LAUNCH(L_, "cconv_bwd", k_cconv_bwd<D>, dim3(grid), dim3(256), 0, xs, y, w, dy, gxs, dxs, scratch, (int)L, tiles, ntiles, pre, post); break
    switch (dil) {
        FD_CC_BWD(1); FD_CC_BWD(2); FD_CC_BWD(3); FD_CC_BWD(4); FD_CC_BWD(9); FD_CC_BWD(27);
    default: return hipErrorInvalidValue;
    }
#undef FD_CC_BWD
    if (dw || db) FD_LAUNCH(L_, "cconv_reduce", k_cconv_reduce, dim3((PSTRIDE + 31) / 32), dim3(256), 0, (const float *)scratch, grid, (int)PSTRIDE, (int)PSTRIDE, (int)NW, dw, db);
    return hipSuccess;
}

// first_audio_conv / final_conv of the training path.  which: 0 = first (x [B,1,L] -> y [B,32,L]), 1 = final (x [B,32,L] -> y [B,1,L])
static int c7_grid(const Launch &L_, int B, int64_t L) { return (int)std::min<int64_t>((int64_t)B * ((L + C7W - 1) / C7W), 4 * (int64_t)L_.ctx->num_cus); }
size_t conv7_scratch_floats(const Launch &L_, int B, int64_t L) { return (size_t)c7_grid(L_, B, L) * C7P; }

hipError_t conv7_forward(const Launch &L_, int which, const float *x, const float *w, const float *bias, float *y, int B, int64_t L)
{
    if (which == 0) FD_LAUNCH(L_, "conv7_first_fwd", k_c7_first_fwd, dim3((unsigned)((L + 1023) / 1024), B), dim3(256), 0, x, w, bias, y, (int)L);
    else {
        const int tiles = (int)((L + C7W - 1) / C7W);
        FD_LAUNCH(L_, "conv7_final_fwd", k_c7_final_fwd, dim3(B * tiles), dim3(256), 0, x, w, bias, y, (int)L, tiles);
    }
    return hipSuccess;
}

hipError_t conv7_backward(const Launch &L_, int which, const float *x, const float *w, const float *dy, float *dx, float *dw, float *db, int B,
                          int64_t L, float *scratch)
{
    const int tiles = (int)((L + C7W - 1) / C7W), ntiles = B * tiles, grid = c7_grid(L_, B, L);
    if (which == 0) FD_LAUNCH(L_, "conv7_first_bwd", k_c7_bwd<0>, dim3(grid), dim3(256), 0, dy, x, w, dx, scratch, (int)L, tiles, ntiles);
    else FD_LAUNCH(L_, "conv7_final_bwd", k_c7_bwd<1>, dim3(grid), dim3(256), 0, x, dy, w, dx, scratch, (int)L, tiles, ntiles);
    if (dw || db) FD_LAUNCH(L_, "cconv_reduce", k_cconv_reduce, dim3((C7P + 31) / 32), dim3(256), 0, (const float *)scratch, grid, (int)C7P, which == 0 ? (int)C7P : 32 * 7 + 1, 32 * 7, dw, db);
    return hipSuccess;
}

}  // namespace fdk
