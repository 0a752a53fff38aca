// fd_kernels_kconv.hip -- the KernelPredictor's `kernel_conv` for the TRAINING path: Conv1d(64 -> M, k = 3, pad 1)
// with M = 24576 (modules/FastDiff/module/modules.py:315-318,330-331) -- and, since round 4, its `bias_conv` (M = 256, :316-318) and the
// six 64 -> 64 convolutions of its residual stack (:303-314): any M that is a multiple of 32 --, forward and the three gradients, in the reference's own tensor
// layouts (h [B,64,T], weight [M,64,3], bias [M], out [B,M,T]) on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32: training
// keeps fp32 products).  Per block of the network this is the largest matrix product of the training step
// (2 * 24576 * 192 * B*T flops each way: 18.9 GFLOP at the reference's batch of 20 x 100 frames), which eager PyTorch runs as MIOpen
// implicit-GEMM kernels with layout transposes around them (0.64 ms forward, 1.49 ms forward + backward per block on an MI355X).
//
//   out[b,p,t]  = bias[p] + sum_{c,k} W[p,c,k] h[b,c,t+k-1]            rows p, cols t,  K = (c,k) = 192      A = W (registers), B = h window (LDS)
//   dW[p,c,k]   = sum_{b,t} dout[b,p,t] h[b,c,t+k-1]                    rows p, cols (c,k), K = (b,t)         A = dout (global float4), B = h window (LDS)
//   G[b,(c,k),t] = sum_p W[p,c,k] dout[b,p,t];  dh[b,c,t] = sum_k G[b,(c,k),t-k+1]     rows (c,k), cols t, K = p, split over workgroups
//   dbias[p]    = sum_{b,t} dout[b,p,t]                                 (rides in the dW kernel)
// MFMA operand layout (32x32x2): A lane l = A[row l&31][k l>>5], B lane l = B[k l>>5][col l&31], D lane l reg r = D[(r&3) + 8(r>>2) + 4(l>>5)][l&31].
// One k-step s of lane half hi stands for the reduction index 8 (s>>2) + 4 hi + (s&3): a lane then owns 4 consecutive indices per 4
// steps, i.e. one aligned float4 of a row-major operand.
#include <stdint.h>

#include <algorithm>

#include "fd_kernels.h"
#include "fd_frame_order.h"

namespace fdk_kconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float f4c(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ int drow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

constexpr int CI = 64, KS = 3, KK = CI * KS;      // 192 = the reduction length of the forward product
// LDS row strides are compile-time constants (T <= 128), so that every per-step offset is an immediate of the ds_read and costs neither
// a register nor an address instruction: the h window [64][LD], x = t + 1 in [0, LD): LD = 131 = 3 (mod 32) makes the dW kernel's B
// reads -- lane = column (c, tap), address c * LD + tap + t -- fall into 32 different banks; columns behind T + 1 are zero.
constexpr int LD = 131, LDD = 129;

// h[b] (64 x T) into LDS with one zero column each side: hs[c * LD + x] = h[b, c, x - 1], x in [0, T + 1]
__device__ __forceinline__ void stage_h(float *__restrict__ hs, const float *__restrict__ h, int b, int T, int tid)
{
    const int n = CI * (T + 2);
    for (int idx = tid; idx < n; idx += 256) {
        const int c = idx / (T + 2), x = idx - c * (T + 2), t = x - 1;
        hs[c * LD + x] = (t >= 0 && t < T) ? h[((int64_t)b * CI + c) * T + t] : 0.0f;
    }
}
__device__ __forceinline__ void zero_h(float *__restrict__ hs, int tid)      // once per workgroup: the columns no staging writes
{
    for (int idx = tid; idx < CI * LD; idx += 256) hs[idx] = 0.0f;
}

// FRAMES (the "frames" entry points; M = layers x 6144): the rows of a 32-row group are not consecutive rows of W but the 32 rows whose
// coefficients lie next to each other in a frame of the LVC operator (group g of layer y: positions 32 g .. 32 g + 31 of the frame, row
// y * 6144 + row_of<ORDER>(position)), and the tensor on the other side is [B][layers][T][6144].  A lane's weight row is 768 B of its own
// either way, so the gather costs nothing.
template <int ORDER>
__device__ __forceinline__ int frame_row(int p0, int j)      // p0 = 32 x (group index over all layers), j = 0 .. 31
{
    const int layer = p0 / fdk_order::ME, e0 = p0 - layer * fdk_order::ME;
    return layer * fdk_order::ME + fdk_order::row_of<ORDER>(e0 + j);
}

// ---- forward: workgroup = 128 output rows (wave = 32 rows, weights in 96 registers) x a range of utterances ------------------------
// FRAMES: the operands change sides -- A = the h window (lane = t), B = W (lane = row) -- so that the accumulators hold out[t][row]
// with the lanes along the rows: one 128 B line of a frame per store.
template <bool FRAMES>
__global__ void __launch_bounds__(256, 2) k_kc_fwd(const float *__restrict__ h, const float *__restrict__ W, const float *__restrict__ bias,
                                                   float *__restrict__ out, int B, int M, int T, int bchunk)
{
    __shared__ float hs[CI * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    // reduction index of step s in lane half hi: 96 hi + s = c * 3 + tap, i.e. the halves split the input channels (32 each):
    // A = 24 aligned float4 of the lane's weight row, B = hs[(32 hi + s / 3) * LD + s % 3 + t]: one base register + an immediate
    const bool live = p0 < M;      // M a multiple of 32, not necessarily of 128: the last workgroup's waves beyond M only stage and wait
    const int prow = !live ? 0 : FRAMES ? frame_row<fdk_order::ORDER_FWD>(p0, l31) : p0 + l31;      // this lane's row of W
    float4 a[24];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(W + (int64_t)prow * KK) + 24 * hi;
#pragma unroll
        for (int q = 0; q < 24; ++q) a[q] = wp[q];
    }
    float bz[FRAMES ? 1 : 16];
    if constexpr (FRAMES) bz[0] = bias[prow];
    else {
#pragma unroll
        for (int r = 0; r < 16; ++r) bz[r] = bias[live ? p0 + drow(r, hi) : 0];
    }
    const int nct = (T + 31) / 32;                      // <= 4 (T <= 128)
    const float *hb = hs + hi * 32 * LD + l31;
    zero_h(hs, tid);
    for (int b = b0; b < b1; ++b) {
        __syncthreads();
        stage_h(hs, h, b, T, tid);
        __syncthreads();
        if (!live) continue;
        // all column tiles of the utterance first, their stores together at the end: the pieces of an output row (T floats, not a
        // multiple of a cache line) then reach L2 back to back and leave it as whole lines
        f32x16 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = bz[FRAMES ? 0 : r];
#ifdef FD_KC_FWD_TILE_MAJOR      // (the first form: one column tile after the other, 96 dependent matrix instructions in a row)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct < nct) {
#pragma unroll
                for (int s = 0; s < 96; ++s) {
                    const float wv = f4c(a[s >> 2], s & 3), hv = hb[(s / 3) * LD + (s % 3) + ct * 32];
                    acc[ct] = FRAMES ? mfma32(hv, wv, acc[ct]) : mfma32(wv, hv, acc[ct]);
                }
            }
        }
#else
        // step-major: the (up to) four column tiles of a k-step back to back -- independent accumulators, one weight register
#pragma unroll
        for (int s = 0; s < 96; ++s) {            // h[c, t + tap - 1] = hs[c][t + tap]; columns behind the utterance read zeros
            const float wv = f4c(a[s >> 2], s & 3);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (ct < nct) {
                    const float hv = hb[(s / 3) * LD + (s % 3) + ct * 32];
                    acc[ct] = FRAMES ? mfma32(hv, wv, acc[ct]) : mfma32(wv, hv, acc[ct]);
                }
            }
        }
#endif
        if constexpr (FRAMES) {      // acc[ct][r] = out[t = 32 ct + drow(r, hi)][row of lane l31]
            // address = (wave-uniform frame base + a constant per (ct, r)) + a 32-bit lane offset: no address registers per store
            const int layer = p0 / fdk_order::ME, e0 = p0 - layer * fdk_order::ME, nl = M / fdk_order::ME;
            float *fu = out + ((int64_t)(b * nl + layer) * T) * fdk_order::ME + e0;
            const int lane_off = 4 * hi * fdk_order::ME + l31;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tu = ct * 32 + (r & 3) + 8 * (r >> 2);      // t = tu + 4 hi
                    if (ct < nct && tu + 4 * hi < T) fu[tu * fdk_order::ME + lane_off] = acc[ct][r];
                }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *orow = out + ((int64_t)b * M + p0 + drow(r, hi)) * T;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    if (ct < nct && ct * 32 + l31 < T) orow[ct * 32 + l31] = acc[ct][r];
            }
        }
    }
}

// ---- dW and dbias: workgroup = 128 rows p (wave = 32 rows), all 192 columns (c, k) in six accumulator tiles, reduction over every
//      (b, t); the A operand (dout) comes straight from global memory, one float4 = four reduction steps per lane -------------------
// FRAMES: dout is [B][layers][T][6144] in the dK accumulator order; a lane's row is then 6144 floats apart from t to t + 1 (one dword
// per step and lane, 128 B per half-wave) instead of one float4 per four steps.
template <bool ALIGNED, bool FRAMES>
__global__ void __launch_bounds__(256, 2) k_kc_dw(const float *__restrict__ h, const float *__restrict__ dout, float *__restrict__ part,
                                                  int B, int M, int T, int bchunk)
{
    // blockIdx.y = a range of utterances: the partial sums of the ranges are added by k_kc_dw_sum in a fixed order (bit-reproducible,
    // unlike atomics); the ranges give the CUs equal loads and let one workgroup load while its neighbour multiplies
    __shared__ float hs[CI * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    f32x16 acc[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
    // reduction index of step (q, j) in lane half hi: t = 8 q + 4 hi + j.  B = h[c, t + tap - 1] = hs[c * LD + tap + t] for the lane's
    // column (c, tap) = 32 ct + l31 of column tile ct: one base register per tile + the immediate 8 q + j
    const float *hb[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) {
        const int kk = ct * 32 + l31;
        hb[ct] = hs + (kk / 3) * LD + (kk % 3) + 4 * hi;
    }
    float accb = 0.0f;
    const int nq = (T + 7) / 8;                    // <= 16
    const bool live = p0 < M;                      // (M a multiple of 32: see k_kc_fwd)
    const int layer = p0 / fdk_order::ME, e0 = p0 - layer * fdk_order::ME, nl = M / fdk_order::ME;      // (FRAMES)
    zero_h(hs, tid);
    for (int b = b0; b < b1; ++b) {
        // this lane's row of dout: all of it is requested before h is staged, so the loads fly during the staging and the barriers
        float4 dv[16];
        if constexpr (FRAMES) {      // (wave-uniform frame base + a constant per step) + a 32-bit lane offset
            const float *du = dout + ((int64_t)(b * nl + (live ? layer : 0)) * T) * fdk_order::ME + (live ? e0 : 0);
            const int lane_off = 4 * hi * fdk_order::ME + l31;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int t0 = 8 * q + 4 * hi;
                if (q < nq)
                    dv[q] = make_float4(t0 < T ? du[(8 * q) * fdk_order::ME + lane_off] : 0.0f, t0 + 1 < T ? du[(8 * q + 1) * fdk_order::ME + lane_off] : 0.0f,
                                        t0 + 2 < T ? du[(8 * q + 2) * fdk_order::ME + lane_off] : 0.0f, t0 + 3 < T ? du[(8 * q + 3) * fdk_order::ME + lane_off] : 0.0f);
            }
        } else {
            const float *dr = dout + ((int64_t)b * M + (live ? p0 + l31 : 0)) * T;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int t0 = 8 * q + 4 * hi;
                if (q < nq) {
                    if (ALIGNED && t0 + 3 < T) dv[q] = *reinterpret_cast<const float4 *>(dr + t0);
                    else dv[q] = make_float4(t0 < T ? dr[t0] : 0.0f, t0 + 1 < T ? dr[t0 + 1] : 0.0f, t0 + 2 < T ? dr[t0 + 2] : 0.0f, t0 + 3 < T ? dr[t0 + 3] : 0.0f);
                }
            }
        }
        __syncthreads();
        stage_h(hs, h, b, T, tid);
        __syncthreads();
        if (!live) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q < nq) {
                accb += (dv[q].x + dv[q].y) + (dv[q].z + dv[q].w);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float av = f4c(dv[q], j);                     // 0 beyond T; the window reads zeros there
#pragma unroll
                    for (int ct = 0; ct < 6; ++ct) acc[ct] = mfma32(av, hb[ct][8 * q + j], acc[ct]);
                }
            }
        }
    }
    if (!live) return;
    // partial sums of this utterance range: [range][M][192] and, behind them, [range][M] for the bias
    float *pw = part + (int64_t)blockIdx.y * M * KK;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int prow = FRAMES ? frame_row<fdk_order::ORDER_DK>(p0, drow(r, hi)) : p0 + drow(r, hi);
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pw[(int64_t)prow * KK + ct * 32 + l31] = acc[ct][r];
    }
    accb += __shfl_xor(accb, 32, 64);
    if (hi == 0) part[(int64_t)gridDim.y * M * KK + (int64_t)blockIdx.y * M + (FRAMES ? frame_row<fdk_order::ORDER_DK>(p0, l31) : p0 + l31)] = accb;
}

// pointer tables that travel as kernel arguments (up to KCS_MULTI convolutions of one shape per launch: blockIdx.z / blockIdx.y)
constexpr int KCS_MULTI = 8;
// n > 0: the kernel takes its three input pointers and its output pointer from slot blockIdx.z (k_kc_dh_fold: blockIdx.y) instead of
// its own arguments: the same launch then serves n independent convolutions of one shape -- the three KernelPredictors of the network
// have identical front ends (input convolution + residual stack) on different weights, each a chain of latency-bound launches of B
// workgroups; side by side they are the same chain with 3 B.
struct KcMulti {
    int n;
    const float *a[KCS_MULTI], *b[KCS_MULTI], *c[KCS_MULTI];
    float *o[KCS_MULTI];
};
struct KcsItems {
    const float *h[KCS_MULTI], *dout[KCS_MULTI], *y[KCS_MULTI];
    float *part[KCS_MULTI];
};
struct KcsSums {
    const float *part[KCS_MULTI];
    float *dW[KCS_MULTI], *dbias[KCS_MULTI];
};

// the utterance ranges added up in a fixed order: dW [M][192] and dbias [M] (either may be null)
__global__ void __launch_bounds__(256) k_kc_dw_sum(const KcsSums sums, int M, int ny, int kk)      // kk: columns of dW (192; the input convolution: 400)
{
    const float *__restrict__ part = sums.part[blockIdx.y];
    float *__restrict__ dW = sums.dW[blockIdx.y], *__restrict__ dbias = sums.dbias[blockIdx.y];
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, nW = (int64_t)M * kk;
    if (idx < nW) {
        if (!dW) return;
        float v = 0.0f;
        for (int y = 0; y < ny; ++y) v += part[y * nW + idx];
        dW[idx] = v;
    } else if (idx < nW + M) {
        if (!dbias) return;
        float v = 0.0f;
        for (int y = 0; y < ny; ++y) v += part[ny * nW + (int64_t)y * M + (idx - nW)];
        dbias[idx - nW] = v;
    }
}

// ---- small M (the predictor's 64 -> 64 convolutions and bias_conv, M = 64 / 256): the kernels above put 128 rows on a workgroup and
//      walk the utterances one after the other -- with one or two row groups in all that leaves the chip empty (48 / 78 us per launch
//      at M = 64).  Here a workgroup is (utterance, 64 rows): wave = (32-row tile, half of the column tiles), B * M / 64 workgroups.
//      post: slope of a leaky-relu on the output (1 = none): the predictor follows each of these convolutions with LeakyReLU(0.1)
//      (modules.py:296-314); the backward kernels then take the activated output y and scale dout by (y > 0 ? 1 : post) as they load it.
__global__ void __launch_bounds__(256, 2) k_kcs_fwd(const float *__restrict__ h, const float *__restrict__ W, const float *__restrict__ bias,
                                                    float *__restrict__ out, int B, int M, int T, float post, const KcMulti m)
{
    __shared__ float hs[CI * LD];
    if (m.n) { h = m.a[blockIdx.z]; W = m.b[blockIdx.z]; bias = m.c[blockIdx.z]; out = m.o[blockIdx.z]; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x, p0 = (blockIdx.y * 2 + (wave & 1)) * 32, ch = wave >> 1;      // ch: column tiles 2 ch, 2 ch + 1
    const bool live = p0 < M;
    float4 a[24];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(W + (int64_t)(live ? p0 + l31 : 0) * KK) + 24 * hi;
#pragma unroll
        for (int q = 0; q < 24; ++q) a[q] = wp[q];
    }
    float bz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bz[r] = bias[live ? p0 + drow(r, hi) : 0];
    zero_h(hs, tid);
    __syncthreads();
    stage_h(hs, h, b, T, tid);
    __syncthreads();
    if (!live) return;
    const float *hb = hs + hi * 32 * LD + l31;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        const int ct = 2 * ch + c2;
        if (ct * 32 >= T) break;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bz[r];
#pragma unroll
        for (int s = 0; s < 96; ++s) acc = mfma32(f4c(a[s >> 2], s & 3), hb[(s / 3) * LD + (s % 3) + ct * 32], acc);
        if (ct * 32 + l31 < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((int64_t)b * M + p0 + drow(r, hi)) * T + ct * 32 + l31] = acc[r] > 0.0f ? acc[r] : acc[r] * post;
        }
    }
}

// dW / dbias partial of ONE utterance: part [B][M][192] and, behind it, [B][M]; k_kc_dw_sum adds the utterances in order.
// wave = (32-row tile, three of the six column tiles (c, k))
// Up to KCS_MULTI convolutions of the same shape in one launch (blockIdx.z; the pointers travel as kernel arguments): the six pairs of
// the predictor's residual stack have their weight gradients computed together once the dx chain has run -- one launch of 120
// workgroups instead of six latency-bound launches of 20.  y[z] = null: dout[z] is already the gradient in front of the activation.

template <bool ALIGNED, bool ACT>
__global__ void __launch_bounds__(256, 2) k_kcs_dw(const KcsItems items, int B, int M, int T, float post)
{
    __shared__ float hs[CI * LD];
    const float *__restrict__ h = items.h[blockIdx.z], *__restrict__ dout = items.dout[blockIdx.z], *__restrict__ y = items.y[blockIdx.z];
    float *__restrict__ part = items.part[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x, p0 = (blockIdx.y * 2 + (wave & 1)) * 32, ch = wave >> 1;      // ch: column tiles 3 ch .. 3 ch + 2
    const bool live = p0 < M;
    const float *dr = dout + ((int64_t)b * M + (live ? p0 + l31 : 0)) * T;
    const int nq = (T + 7) / 8;
    float4 dv[16];
    auto row4 = [&](const float *r, int t0) {
        if (ALIGNED && t0 + 3 < T) return *reinterpret_cast<const float4 *>(r + t0);
        return make_float4(t0 < T ? r[t0] : 0.0f, t0 + 1 < T ? r[t0 + 1] : 0.0f, t0 + 2 < T ? r[t0 + 2] : 0.0f, t0 + 3 < T ? r[t0 + 3] : 0.0f);
    };
    float4 yv[ACT ? 16 : 1];      // ACT: the activated output, requested together with dout (one round trip for both)
    const float *yr = (ACT && y) ? y + ((int64_t)b * M + (live ? p0 + l31 : 0)) * T : nullptr;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (q < nq) {
            dv[q] = row4(dr, 8 * q + 4 * hi);
            if constexpr (ACT) yv[q] = yr ? row4(yr, 8 * q + 4 * hi) : make_float4(1.f, 1.f, 1.f, 1.f);      // (no y: mask = 1)
        }
    zero_h(hs, tid);
    __syncthreads();
    stage_h(hs, h, b, T, tid);
    __syncthreads();
    if (!live) return;
    if constexpr (ACT) {      // the gradient in front of the fused activation
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q < nq)
                dv[q] = make_float4(yv[q].x > 0.0f ? dv[q].x : dv[q].x * post, yv[q].y > 0.0f ? dv[q].y : dv[q].y * post,
                                    yv[q].z > 0.0f ? dv[q].z : dv[q].z * post, yv[q].w > 0.0f ? dv[q].w : dv[q].w * post);
    }
    f32x16 acc[3];
    const float *hb[3];
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c3][r] = 0.0f;
        const int kk = (3 * ch + c3) * 32 + l31;
        hb[c3] = hs + (kk / 3) * LD + (kk % 3) + 4 * hi;
    }
    float accb = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (q < nq) {
            accb += (dv[q].x + dv[q].y) + (dv[q].z + dv[q].w);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float av = f4c(dv[q], j);
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) acc[c3] = mfma32(av, hb[c3][8 * q + j], acc[c3]);
            }
        }
    }
    float *pw = part + (int64_t)b * M * KK;
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3)
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[(int64_t)(p0 + drow(r, hi)) * KK + (3 * ch + c3) * 32 + l31] = acc[c3][r];
    accb += __shfl_xor(accb, 32, 64);
    if (hi == 0 && ch == 0) part[(int64_t)B * M * KK + (int64_t)b * M + p0 + l31] = accb;
}

// ---- dh, first pass: workgroup = (slice of the rows p, utterance b): G_part[(c,k), t] = sum over the slice of W[p,(c,k)] dout[b,p,t];
//      wave = one 32-column tile of t (T <= 128), all six 32-row tiles of (c,k); W and dout go through LDS 32 rows at a time ----------
// FRAMES: a chunk is one 32-row group of a layer's frame order: its rows of W are gathered (768 B each), its piece of dout is 32
// consecutive floats of every frame of the utterance.
template <bool FRAMES, bool ACT = false>
__global__ void __launch_bounds__(256, 2) k_kc_dh(const float *__restrict__ W, const float *__restrict__ dout, float *__restrict__ part,
                                                  int B, int M, int T, int prows, const float *__restrict__ y, float post, const KcMulti m)
{
    if (m.n) { W = m.a[blockIdx.z]; dout = m.b[blockIdx.z]; y = m.c[blockIdx.z]; part = m.o[blockIdx.z]; }
    __shared__ __attribute__((aligned(16))) float ws[32 * KK];
    __shared__ float dsm[32 * LDD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int ks = blockIdx.x, b = blockIdx.y, pbeg = ks * prows;
    const int tcol = wave * 32 + l31, tc = min(tcol, T - 1);
    const bool wave_live = wave * 32 < T;
    const float *wb = ws + hi * KK + l31, *db = dsm + hi * LDD + tc;      // lane bases: every per-step offset below is an immediate
    f32x16 acc[6];
#pragma unroll
    for (int rt = 0; rt < 6; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
    // 32 rows of W (6 float4 per thread) and of dout (<= 16 floats per thread) per chunk; the next chunk's loads are in flight under
    // the current chunk's matrix work
    typedef float f4n __attribute__((ext_vector_type(4)));
    f4n wv[6];      // (a native vector type: the HIP float4 struct kept this prefetch buffer in scratch)
    float dvv[16], yvv[ACT ? 16 : 1];      // ACT: the activated output next to dout (k_kcs_fwd)
    const int nd = (32 * T + 255) / 256;          // <= 16
    auto load_chunk = [&](int pc) {
        if constexpr (FRAMES) {
            const int layer = pc / fdk_order::ME, e0 = pc - layer * fdk_order::ME, nl = M / fdk_order::ME;
#pragma unroll
            for (int k = 0; k < 6; ++k) {      // row j of the chunk = 48 float4
                const int idx = k * 256 + tid, j = idx / 48, c4 = idx - j * 48;
                wv[k] = reinterpret_cast<const f4n *>(W + (int64_t)frame_row<fdk_order::ORDER_DK>(pc, j) * KK)[c4];
            }
            const float *dr = dout + ((int64_t)(b * nl + layer) * T) * fdk_order::ME + e0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {      // (t, j): 32 consecutive floats per frame
                const int idx = k * 256 + tid;
                if (k < nd) dvv[k] = idx < 32 * T ? dr[(int64_t)(idx >> 5) * fdk_order::ME + (idx & 31)] : 0.0f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) wv[k] = reinterpret_cast<const f4n *>(W + (int64_t)pc * KK)[k * 256 + tid];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int idx = k * 256 + tid;
                if (k < nd) {
                    const int64_t at = ((int64_t)b * M + pc + idx / T) * T + idx % T;
                    dvv[k] = idx < 32 * T ? dout[at] : 0.0f;
                    if constexpr (ACT) yvv[k] = (y && idx < 32 * T) ? y[at] : 1.0f;      // (no y for this item: mask = 1)
                }
            }
        }
    };
    load_chunk(pbeg);
    for (int pc = pbeg; pc < pbeg + prows; pc += 32) {
        __syncthreads();                          // the previous chunk's reads are done
#pragma unroll
        for (int k = 0; k < 6; ++k) reinterpret_cast<f4n *>(ws)[k * 256 + tid] = wv[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = k * 256 + tid;
            if (k < nd && idx < 32 * T) {
                if constexpr (FRAMES) dsm[(idx & 31) * LDD + (idx >> 5)] = dvv[k];
                else if constexpr (ACT) dsm[(idx / T) * LDD + idx % T] = yvv[k] > 0.0f ? dvv[k] : dvv[k] * post;      // the gradient in front of the activation
                else dsm[(idx / T) * LDD + idx % T] = dvv[k];
            }
        }
        __syncthreads();
        if (pc + 32 < pbeg + prows) load_chunk(pc + 32);
        if (wave_live) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float bv = db[2 * s * LDD];                        // row p = 2 s + hi of the chunk = the reduction index
#pragma unroll
                for (int rt = 0; rt < 6; ++rt) acc[rt] = mfma32(wb[2 * s * KK + rt * 32], bv, acc[rt]);
            }
        }
    }
    if (wave_live && tcol < T) {
        float *g = part + ((int64_t)ks * B + b) * KK * T;
#pragma unroll
        for (int rt = 0; rt < 6; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[(int64_t)(rt * 32 + drow(r, hi)) * T + tcol] = acc[rt][r];
    }
}

// ---- dh, second pass: add the slices and fold the three taps: dh[b,c,t] = sum_ks sum_k G[ks][b][(c,k)][t - k + 1] --------------------
//      hin / in_slope (a chain of "Conv1d, LeakyReLU" pairs: the predictor's residual stack): h is itself the activated output of the
//      pair below, and dh is wanted in front of THAT activation: dh *= (h > 0 ? 1 : in_slope) on the way out, so the pair below runs
//      its backward on plain kernels (no mask loads in its two latency-bound launches).  hin = null: dh as it is.
__global__ void __launch_bounds__(256) k_kc_dh_fold(const float *__restrict__ part, float *__restrict__ dh, int B, int T, int nks,
                                                    const float *__restrict__ hin, float in_slope, const KcMulti m)
{
    if (m.n) { part = m.a[blockIdx.y]; hin = m.b[blockIdx.y]; dh = m.o[blockIdx.y]; }
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * CI * T) return;
    const int t = idx % T, c = (idx / T) % CI, b = idx / (T * CI);
    float v = 0.0f;
    for (int ks = 0; ks < nks; ++ks) {
        const float *g = part + (((int64_t)ks * B + b) * KK + c * KS) * T;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int tq = t - k + 1;
            if (tq >= 0 && tq < T) v += g[(int64_t)k * T + tq];
        }
    }
    dh[idx] = (hin && !(hin[idx] > 0.0f)) ? v * in_slope : v;
}

// ---- the predictor's input convolution with its activation: Conv1d(80 -> 64, k5, pad 2), LeakyReLU (modules.py:292-295) -----------
//      0.1 GFLOP at the training shape: plain VALU through LDS, workgroup = (utterance, 16 channels), thread = (channel, 8 frames).
//      xs / gs rows: [2 zeros][T values][zeros up to IC_LD], so that a thread's 12-float window is three aligned ds_read_b128.
constexpr int IC_CI = 80, IC_CO = 64, IC_K = 5, IC_KK = IC_CI * IC_K, IC_LD = 136;

__device__ __forceinline__ void ic_stage_x(float *__restrict__ xs, const float *__restrict__ x, int b, int T, int tid)
{
    for (int idx = tid; idx < IC_CI * IC_LD; idx += 256) {
        const int c = idx / IC_LD, t = idx - c * IC_LD - 2;
        xs[idx] = (t >= 0 && t < T) ? x[((int64_t)b * IC_CI + c) * T + t] : 0.0f;
    }
}
// the gradient in front of the activation: dy * (y > 0 ? 1 : post), rows o0 .. o0 + rows - 1
__device__ __forceinline__ void ic_stage_g(float *__restrict__ gs, const float *__restrict__ dy, const float *__restrict__ y, int b, int o0, int rows,
                                           int T, float post, int tid)
{
    for (int idx = tid; idx < rows * IC_LD; idx += 256) {
        const int o = idx / IC_LD, t = idx - o * IC_LD - 2;
        float v = 0.0f;
        if (t >= 0 && t < T) {
            const int64_t at = ((int64_t)b * IC_CO + o0 + o) * T + t;
            v = y[at] > 0.0f ? dy[at] : dy[at] * post;
        }
        gs[idx] = v;
    }
}

__global__ void __launch_bounds__(256) k_ic_fwd(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                float *__restrict__ out, int T, float post, const KcMulti m)
{
    __shared__ __attribute__((aligned(16))) float xs[IC_CI * IC_LD];
    if (m.n) { x = m.a[blockIdx.z]; w = m.b[blockIdx.z]; bias = m.c[blockIdx.z]; out = m.o[blockIdx.z]; }
    const int b = blockIdx.x, tid = threadIdx.x, o = blockIdx.y * 16 + (tid >> 4), t0 = (tid & 15) * 8;
    ic_stage_x(xs, x, b, T, tid);
    __syncthreads();
    if (t0 >= T) return;
    float acc[8];
    const float bv = bias[o];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bv;
    const float *wr = w + (int64_t)o * IC_KK;
    for (int c = 0; c < IC_CI; ++c) {
        float win[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(xs + c * IC_LD + t0 + 4 * q);
            win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < IC_K; ++k) {      // x[t + k - 2] = xs[t + k]
            const float wv = wr[c * IC_K + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(wv, win[j + k], acc[j]);
        }
    }
    float *orow = out + ((int64_t)b * IC_CO + o) * T + t0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (t0 + j < T) orow[j] = acc[j] > 0.0f ? acc[j] : acc[j] * post;
}

// dx[b, c, t] = sum_{o, k} w[o, c, k] g[b, o, t + 2 - k]: workgroup = (utterance, 16 input channels)
__global__ void __launch_bounds__(256) k_ic_bwd_x(const float *__restrict__ w, const float *__restrict__ dy, const float *__restrict__ y,
                                                  float *__restrict__ dx, int T, float post, const KcMulti m)
{
    __shared__ __attribute__((aligned(16))) float gs[IC_CO * IC_LD];
    if (m.n) { w = m.a[blockIdx.z]; dy = m.b[blockIdx.z]; y = m.c[blockIdx.z]; dx = m.o[blockIdx.z]; }
    const int b = blockIdx.x, tid = threadIdx.x, c = blockIdx.y * 16 + (tid >> 4), t0 = (tid & 15) * 8;
    ic_stage_g(gs, dy, y, b, 0, IC_CO, T, post, tid);
    __syncthreads();
    if (t0 >= T) return;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    for (int o = 0; o < IC_CO; ++o) {
        float win[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(gs + o * IC_LD + t0 + 4 * q);
            win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
        }
        const float *wr = w + (int64_t)o * IC_KK + c * IC_K;
#pragma unroll
        for (int k = 0; k < IC_K; ++k) {      // g[t + 2 - k] = gs[t + 4 - k]
            const float wv = wr[k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(wv, win[j + 4 - k], acc[j]);
        }
    }
    float *drow = dx + ((int64_t)b * IC_CI + c) * T + t0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (t0 + j < T) drow[j] = acc[j];
}

// this utterance's share of dW[o, c, k] = sum_t g[o, t] x[c, t + k - 2] and of db[o] = sum_t g[o, t]: part [B][64][400], then [B][64]
// (k_kc_dw_sum adds the utterances in order).  workgroup = (utterance, 16 output channels), thread = (channel o, 5 input channels)
__global__ void __launch_bounds__(256) k_ic_bwd_w(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ y,
                                                  float *__restrict__ part, int B, int T, float post, const KcMulti m)
{
    __shared__ __attribute__((aligned(16))) float xs[IC_CI * IC_LD];
    if (m.n) { x = m.a[blockIdx.z]; dy = m.b[blockIdx.z]; y = m.c[blockIdx.z]; part = m.o[blockIdx.z]; }
    __shared__ __attribute__((aligned(16))) float gs[16 * IC_LD];
    const int b = blockIdx.x, tid = threadIdx.x, ol = tid >> 4, o = blockIdx.y * 16 + ol, c0 = (tid & 15) * 5;
    ic_stage_x(xs, x, b, T, tid);
    ic_stage_g(gs, dy, y, b, blockIdx.y * 16, 16, T, post, tid);
    __syncthreads();
    float acc[5][IC_K];
#pragma unroll
    for (int ci = 0; ci < 5; ++ci)
#pragma unroll
        for (int k = 0; k < IC_K; ++k) acc[ci][k] = 0.0f;
    float accb = 0.0f;
    for (int t = 0; t < T; t += 4) {      // (rows are zero behind T)
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = gs[ol * IC_LD + 2 + t + j];      // g[t + j] (the row starts 8 B into a 16 B slot: scalars)
        accb += (g[0] + g[1]) + (g[2] + g[3]);
#pragma unroll
        for (int ci = 0; ci < 5; ++ci) {
            float xw[8];      // x[t + j + k - 2] = xs[t + j + k], j < 4, k < 5
            const float4 v0 = *reinterpret_cast<const float4 *>(xs + (c0 + ci) * IC_LD + t), v1 = *reinterpret_cast<const float4 *>(xs + (c0 + ci) * IC_LD + t + 4);
            xw[0] = v0.x; xw[1] = v0.y; xw[2] = v0.z; xw[3] = v0.w; xw[4] = v1.x; xw[5] = v1.y; xw[6] = v1.z; xw[7] = v1.w;
#pragma unroll
            for (int k = 0; k < IC_K; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ci][k] = fmaf(g[j], xw[j + k], acc[ci][k]);
        }
    }
    float *pw = part + ((int64_t)b * IC_CO + o) * IC_KK + c0 * IC_K;
#pragma unroll
    for (int ci = 0; ci < 5; ++ci)
#pragma unroll
        for (int k = 0; k < IC_K; ++k) pw[ci * IC_K + k] = acc[ci][k];
    if ((tid & 15) == 0) part[(int64_t)B * IC_CO * IC_KK + (int64_t)b * IC_CO + o] = accb;
}

}  // namespace fdk_kconv

namespace fdk {
using namespace fdk_kconv;

bool kconv_supported(int M, int T) { return M > 0 && M % 32 == 0 && T >= 1 && T <= 128; }
bool kconv_frames_supported(int M, int T) { return kconv_supported(M, T) && M % fdk_order::ME == 0 && M > 512; }

// How many ranges to cut `units` (utterances / row chunks) into when `per_range` workgroups work on each range: the launch lasts
// (rounds of workgroups over the chip's slots) x (units per range); the smallest product wins, then the fewest ranges.  slots: the
// workgroups the chip holds at a time -- TWO per CU for the kernels below (195-226 registers, 33-41 KB of LDS).  Round 4 found the model
// counting one per CU: kernel_conv at the training shape then got 768 workgroups of 5 utterances = two rounds of 5 on 512 slots where
// 960 workgroups of 4 are two rounds of 4 (forward and dW -20 %), and the dh pass 1280 workgroups (2.5 rounds) where 24 slices fill one.
static int pick_ranges(int per_range, int units, int max_ranges, int slots)
{
    int best = 1;
    int64_t best_cost = INT64_MAX;
    for (int n = 1; n <= std::min(units, max_ranges); ++n) {
        const int64_t cost = (int64_t)(((int64_t)per_range * n + slots - 1) / slots) * ((units + n - 1) / n);
        if (cost < best_cost) { best_cost = cost; best = n; }
    }
    return best;
}
// the dh pass: slices of whole 32-row chunks (any divisor of their number up to KC_DH_SLICES), `per_slice` workgroups each
static int dh_slices(int M, int per_slice, int slots)
{
    const int chunks = M / 32;
    int nks = 1;
    int64_t best = INT64_MAX;
    for (int k = 1; k <= 64 && k <= chunks; ++k) {
        if (chunks % k) continue;
        const int64_t cost = (int64_t)(((int64_t)k * per_slice + slots - 1) / slots) * (chunks / k);
        if (cost < best) { best = cost; nks = k; }
    }
    return nks;
}
constexpr int KC_DW_RANGES = 8, KC_DH_SLICES = 64;
// scratch: the dh pass's row slices [slices][B][192][T], then the dW pass's utterance ranges [ranges][M][192] + [ranges][M]
constexpr int KC_SMALL_M = 512;      // up to here a workgroup is (utterance, 64 rows): k_kcs_fwd / k_kcs_dw
size_t kconv_scratch_floats(int B, int M, int T)
{
    return (size_t)KC_DH_SLICES * B * KK * T + (size_t)(M <= KC_SMALL_M ? std::max(B, KC_DW_RANGES) : KC_DW_RANGES) * M * (KK + 1);
}

bool kconv_act_supported(int M, int T) { return kconv_supported(M, T) && M <= KC_SMALL_M; }

hipError_t kconv_forward(const Launch &L, const float *h, const float *W, const float *bias, float *out, int B, int M, int T, bool frames, float post)
{
    if (frames && !kconv_frames_supported(M, T)) return hipErrorInvalidValue;
    if (post != 1.0f && !kconv_act_supported(M, T)) return hipErrorInvalidValue;
    if (M <= KC_SMALL_M) {
        FD_LAUNCH(L, "kconv_forward_small", k_kcs_fwd, dim3(B, (M + 63) / 64), dim3(256), 0, h, W, bias, out, B, M, T, post, KcMulti{});
        return hipSuccess;
    }
    const int gx = (M + 127) / 128;
    const int ny0 = pick_ranges(gx, B, 16, 2 * L.ctx->num_cus), bchunk = (B + ny0 - 1) / ny0, ny = (B + bchunk - 1) / bchunk;
    if (frames) FD_LAUNCH(L, "kconv_forward", k_kc_fwd<true>, dim3(gx, ny), dim3(256), 0, h, W, bias, out, B, M, T, bchunk);
    else FD_LAUNCH(L, "kconv_forward", k_kc_fwd<false>, dim3(gx, ny), dim3(256), 0, h, W, bias, out, B, M, T, bchunk);
    return hipSuccess;
}

size_t input_conv_scratch_floats(int B) { return (size_t)B * IC_CO * (IC_KK + 1); }

hipError_t input_conv_forward(const Launch &L, const float *x, const float *w, const float *bias, float *out, int B, int T, float post)
{
    FD_LAUNCH(L, "input_conv_forward", k_ic_fwd, dim3(B, IC_CO / 16), dim3(256), 0, x, w, bias, out, T, post, KcMulti{});
    return hipSuccess;
}

static KcsSums one_sum(const float *part, float *dW, float *dbias)
{
    KcsSums s = {};
    s.part[0] = part; s.dW[0] = dW; s.dbias[0] = dbias;
    return s;
}

hipError_t input_conv_backward(const Launch &L, const float *x, const float *w, const float *y, const float *dy, float *dx, float *dw, float *db, int B,
                               int T, float post, float *scratch)
{
    if (dx) FD_LAUNCH(L, "input_conv_backward_x", k_ic_bwd_x, dim3(B, IC_CI / 16), dim3(256), 0, w, dy, y, dx, T, post, KcMulti{});
    if (dw || db) {
        FD_LAUNCH(L, "input_conv_backward_w", k_ic_bwd_w, dim3(B, IC_CO / 16), dim3(256), 0, x, dy, y, scratch, B, T, post, KcMulti{});
        FD_LAUNCH(L, "kconv_backward_w_sum", k_kc_dw_sum, dim3((unsigned)(((int64_t)IC_CO * (IC_KK + 1) + 255) / 256)), dim3(256), 0, one_sum(scratch, dw, db),
                  IC_CO, B, IC_KK);
    }
    return hipSuccess;
}

// ---- n <= KCS_MULTI independent convolutions of one shape side by side (KcMulti) -------------------------------------------------
hipError_t kconv_forward_multi(const Launch &L, int n, const float *const *h, const float *const *W, const float *const *bias, float *const *out, int B,
                               int M, int T, float post)
{
    if (n < 1 || n > KCS_MULTI || !kconv_act_supported(M, T)) return hipErrorInvalidValue;
    KcMulti m = {};
    m.n = n;
    for (int i = 0; i < n; ++i) { m.a[i] = h[i]; m.b[i] = W[i]; m.c[i] = bias[i]; m.o[i] = out[i]; }
    FD_LAUNCH(L, "kconv_forward_small", k_kcs_fwd, dim3(B, (M + 63) / 64, n), dim3(256), 0, h[0], W[0], bias[0], out[0], B, M, T, post, m);
    return hipSuccess;
}

size_t kconv_x_multi_scratch_floats(int n, int B, int M, int T) { return (size_t)n * KC_DH_SLICES * B * KK * T; }
// dh[i] = conv_transpose(dout[i] masked with y[i] if given) (* mask of h[i] when in_slope != 1): one step of n dx chains
hipError_t kconv_backward_x_multi(const Launch &L, int n, const float *const *h, const float *const *W, const float *const *y, const float *const *dout,
                                  float *const *dh, int B, int M, int T, float post, float in_slope, float *scratch)
{
    if (n < 1 || n > KCS_MULTI || !kconv_act_supported(M, T)) return hipErrorInvalidValue;
    const int nks = dh_slices(M, B, 2 * L.ctx->num_cus);      // as for ONE convolution (kconv_backward): the same partial sums, the same bits
    KcMulti m = {}, f = {};
    m.n = f.n = n;
    bool any_y = false;
    for (int i = 0; i < n; ++i) {
        float *part = scratch + (size_t)i * nks * B * KK * T;
        m.a[i] = W[i]; m.b[i] = dout[i]; m.c[i] = y ? y[i] : nullptr; m.o[i] = part;
        f.a[i] = part; f.b[i] = in_slope != 1.0f ? h[i] : nullptr; f.o[i] = dh[i];
        any_y = any_y || m.c[i];
    }
    if (any_y) FD_LAUNCH(L, "kconv_backward_h", (k_kc_dh<false, true>), dim3(nks, B, n), dim3(256), 0, W[0], dout[0], m.o[0], B, M, T, M / nks, m.c[0], post, m);
    else FD_LAUNCH(L, "kconv_backward_h", (k_kc_dh<false, false>), dim3(nks, B, n), dim3(256), 0, W[0], dout[0], m.o[0], B, M, T, M / nks, m.c[0], post, m);
    FD_LAUNCH(L, "kconv_backward_h_fold", k_kc_dh_fold, dim3((B * CI * T + 255) / 256, n), dim3(256), 0, f.a[0], dh[0], B, T, nks, f.b[0], in_slope, f);
    return hipSuccess;
}

hipError_t input_conv_forward_multi(const Launch &L, int n, const float *const *x, const float *const *w, const float *const *bias, float *const *out,
                                    int B, int T, float post)
{
    if (n < 1 || n > KCS_MULTI) return hipErrorInvalidValue;
    KcMulti m = {};
    m.n = n;
    for (int i = 0; i < n; ++i) { m.a[i] = x[i]; m.b[i] = w[i]; m.c[i] = bias[i]; m.o[i] = out[i]; }
    FD_LAUNCH(L, "input_conv_forward", k_ic_fwd, dim3(B, IC_CO / 16, n), dim3(256), 0, x[0], w[0], bias[0], out[0], T, post, m);
    return hipSuccess;
}

size_t input_conv_multi_scratch_floats(int n, int B) { return (size_t)n * input_conv_scratch_floats(B); }
hipError_t input_conv_backward_multi(const Launch &L, int n, const float *const *x, const float *const *w, const float *const *y, const float *const *dy,
                                     float *const *dx, float *const *dw, float *const *db, int B, int T, float post, float *scratch)
{
    if (n < 1 || n > KCS_MULTI) return hipErrorInvalidValue;
    if (dx) {
        KcMulti m = {};
        m.n = n;
        for (int i = 0; i < n; ++i) { m.a[i] = w[i]; m.b[i] = dy[i]; m.c[i] = y[i]; m.o[i] = dx[i]; }
        FD_LAUNCH(L, "input_conv_backward_x", k_ic_bwd_x, dim3(B, IC_CI / 16, n), dim3(256), 0, w[0], dy[0], y[0], dx[0], T, post, m);
    }
    if (dw || db) {
        KcMulti m = {};
        KcsSums su = {};
        m.n = n;
        for (int i = 0; i < n; ++i) {
            float *part = scratch + (size_t)i * input_conv_scratch_floats(B);
            m.a[i] = x[i]; m.b[i] = dy[i]; m.c[i] = y[i]; m.o[i] = part;
            su.part[i] = part; su.dW[i] = dw ? dw[i] : nullptr; su.dbias[i] = db ? db[i] : nullptr;
        }
        FD_LAUNCH(L, "input_conv_backward_w", k_ic_bwd_w, dim3(B, IC_CO / 16, n), dim3(256), 0, x[0], dy[0], y[0], m.o[0], B, T, post, m);
        FD_LAUNCH(L, "kconv_backward_w_sum", k_kc_dw_sum, dim3((unsigned)(((int64_t)IC_CO * (IC_KK + 1) + 255) / 256), n), dim3(256), 0, su, IC_CO, B, IC_KK);
    }
    return hipSuccess;
}

// the weight / bias gradients of n <= KCS_MULTI small convolutions of one shape (M <= 512) in two launches; scratch as kconv_backward's
size_t kconv_w_multi_scratch_floats(int n, int B, int M) { return (size_t)n * B * M * (KK + 1); }
hipError_t kconv_backward_w_multi(const Launch &L, int n, const float *const *h, const float *const *dout, const float *const *y, float post, int B,
                                  int M, int T, float *const *dW, float *const *dbias, float *scratch)
{
    if (n < 1 || n > KCS_MULTI || !kconv_act_supported(M, T)) return hipErrorInvalidValue;
    KcsItems it = {};
    KcsSums su = {};
    bool any_y = false;
    for (int i = 0; i < n; ++i) {
        it.h[i] = h[i]; it.dout[i] = dout[i]; it.y[i] = y ? y[i] : nullptr;
        it.part[i] = scratch + (size_t)i * B * M * (KK + 1);
        su.part[i] = it.part[i]; su.dW[i] = dW ? dW[i] : nullptr; su.dbias[i] = dbias ? dbias[i] : nullptr;
        any_y = any_y || it.y[i];
    }
#define FD_KCS_DW(AL_, ACT_) FD_LAUNCH(L, "kconv_backward_w_small", (k_kcs_dw<AL_, ACT_>), dim3(B, (M + 63) / 64, n), dim3(256), 0, it, B, M, T, post)
    if (T % 4 == 0) { if (any_y) FD_KCS_DW(true, true); else FD_KCS_DW(true, false); }
    else { if (any_y) FD_KCS_DW(false, true); else FD_KCS_DW(false, false); }
#undef FD_KCS_DW
    FD_LAUNCH(L, "kconv_backward_w_sum", k_kc_dw_sum, dim3((unsigned)(((int64_t)M * (KK + 1) + 255) / 256), n), dim3(256), 0, su, M, B, KK);
    return hipSuccess;
}

hipError_t kconv_backward(const Launch &L, const float *h, const float *W, const float *dout, float *dh, float *dW, float *dbias, int B, int M,
                          int T, float *scratch, bool frames, const float *y, float post, float in_slope)
{
    if (frames && !kconv_frames_supported(M, T)) return hipErrorInvalidValue;
    if (y && !kconv_act_supported(M, T)) return hipErrorInvalidValue;
    float *part_h = scratch, *part_w = scratch + (size_t)KC_DH_SLICES * B * KK * T;
    if ((dW || dbias) && M <= KC_SMALL_M) {
        KcsItems it = {};
        it.h[0] = h; it.dout[0] = dout; it.y[0] = y; it.part[0] = part_w;
#define FD_KCS_DW(AL_, ACT_) FD_LAUNCH(L, "kconv_backward_w_small", (k_kcs_dw<AL_, ACT_>), dim3(B, (M + 63) / 64), dim3(256), 0, it, B, M, T, post)
        if (T % 4 == 0) { if (y) FD_KCS_DW(true, true); else FD_KCS_DW(true, false); }
        else { if (y) FD_KCS_DW(false, true); else FD_KCS_DW(false, false); }
#undef FD_KCS_DW
        FD_LAUNCH(L, "kconv_backward_w_sum", k_kc_dw_sum, dim3((unsigned)(((int64_t)M * (KK + 1) + 255) / 256)), dim3(256), 0, one_sum(part_w, dW, dbias),
                  M, B, KK);
    } else if (dW || dbias) {
        const int gx = (M + 127) / 128;
        const int ny0 = pick_ranges(gx, B, KC_DW_RANGES, 2 * L.ctx->num_cus), bchunk = (B + ny0 - 1) / ny0, ny = (B + bchunk - 1) / bchunk;
        if (frames) FD_LAUNCH(L, "kconv_backward_w", (k_kc_dw<false, true>), dim3(gx, ny), dim3(256), 0, h, dout, part_w, B, M, T, bchunk);
        else if (T % 4 == 0) FD_LAUNCH(L, "kconv_backward_w", (k_kc_dw<true, false>), dim3(gx, ny), dim3(256), 0, h, dout, part_w, B, M, T, bchunk);
        else FD_LAUNCH(L, "kconv_backward_w", (k_kc_dw<false, false>), dim3(gx, ny), dim3(256), 0, h, dout, part_w, B, M, T, bchunk);
        FD_LAUNCH(L, "kconv_backward_w_sum", k_kc_dw_sum, dim3((unsigned)(((int64_t)M * (KK + 1) + 255) / 256)), dim3(256), 0, one_sum(part_w, dW, dbias),
                  M, ny, KK);
    }
    if (dh) {
        const int nks = dh_slices(M, B, 2 * L.ctx->num_cus);
        if (frames) FD_LAUNCH(L, "kconv_backward_h", (k_kc_dh<true, false>), dim3(nks, B), dim3(256), 0, W, dout, part_h, B, M, T, M / nks, (const float *)nullptr, 1.0f, KcMulti{});
        else if (y) FD_LAUNCH(L, "kconv_backward_h", (k_kc_dh<false, true>), dim3(nks, B), dim3(256), 0, W, dout, part_h, B, M, T, M / nks, y, post, KcMulti{});
        else FD_LAUNCH(L, "kconv_backward_h", (k_kc_dh<false, false>), dim3(nks, B), dim3(256), 0, W, dout, part_h, B, M, T, M / nks, y, post, KcMulti{});
        FD_LAUNCH(L, "kconv_backward_h_fold", k_kc_dh_fold, dim3((B * CI * T + 255) / 256), dim3(256), 0, (const float *)part_h, dh, B, T, nks,
                  in_slope != 1.0f ? h : (const float *)nullptr, in_slope, KcMulti{});
    }
    return hipSuccess;
}

}  // namespace fdk
