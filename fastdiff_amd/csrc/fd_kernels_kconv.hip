// fd_kernels_kconv.hip -- the KernelPredictor's `kernel_conv` for the TRAINING path: Conv1d(64 -> M, k = 3, pad 1)
// with M = 24576 (modules/FastDiff/module/modules.py:315-318,330-331), forward and the three gradients, in the reference's own tensor
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
#include <algorithm>

#include "fd_kernels.h"

namespace fdk_kconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float f4c(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ int drow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

constexpr int CI = 64, KS = 3, KK = CI * KS;      // 192 = the reduction length of the forward product

// h[b] (64 x T) into LDS with one zero column each side: hs[c * LD + x] = h[b, c, x - 1], x in [0, T + 1]
__device__ __forceinline__ void stage_h(float *__restrict__ hs, const float *__restrict__ h, int b, int T, int LD, int tid)
{
    const int n = CI * (T + 2);
    for (int idx = tid; idx < n; idx += 256) {
        const int c = idx / (T + 2), x = idx - c * (T + 2), t = x - 1;
        hs[c * LD + x] = (t >= 0 && t < T) ? h[((int64_t)b * CI + c) * T + t] : 0.0f;
    }
}

// ---- forward: workgroup = 128 output rows (wave = 32 rows, weights in 96 registers) x a range of utterances ------------------------
__global__ void __launch_bounds__(256, 2) k_kc_fwd(const float *__restrict__ h, const float *__restrict__ W, const float *__restrict__ bias,
                                                   float *__restrict__ out, int B, int M, int T, int LD, int bchunk)
{
    extern __shared__ float hs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    float4 a[24];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(W + (int64_t)(p0 + l31) * KK);
#pragma unroll
        for (int q = 0; q < 24; ++q) a[q] = wp[2 * q + hi];
    }
    float bz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bz[r] = bias[p0 + drow(r, hi)];
    const int nct = (T + 31) / 32;
    for (int b = b0; b < b1; ++b) {
        __syncthreads();
        stage_h(hs, h, b, T, LD, tid);
        __syncthreads();
        for (int ct = 0; ct < nct; ++ct) {
            const int tcol = ct * 32 + l31, tc = min(tcol, T - 1);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bz[r];
#pragma unroll
            for (int s = 0; s < 96; ++s) {
                const int k0 = 8 * (s >> 2) + (s & 3), k1 = k0 + 4;                   // reduction index (c, tap) of the two lane halves
                const int o0 = (k0 / 3) * LD + (k0 % 3), o1 = (k1 / 3) * LD + (k1 % 3);
                acc = mfma32(f4c(a[s >> 2], s & 3), hs[(hi ? o1 : o0) + tc], acc);     // h[c, t + tap - 1] = hs[c][t + tap]
            }
            if (tcol < T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[((int64_t)b * M + p0 + drow(r, hi)) * T + tcol] = acc[r];
            }
        }
    }
}

// ---- dW and dbias: workgroup = 128 rows p (wave = 32 rows), all 192 columns (c, k) in six accumulator tiles, reduction over every
//      (b, t); the A operand (dout) comes straight from global memory, one float4 = four reduction steps per lane -------------------
template <bool ALIGNED>
__global__ void __launch_bounds__(256, 2) k_kc_dw(const float *__restrict__ h, const float *__restrict__ dout, float *__restrict__ dW,
                                                  float *__restrict__ dbias, int B, int M, int T, int LD)
{
    extern __shared__ float hs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    f32x16 acc[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
    int offb[6];                                   // this lane's column (c, tap) of every column tile: hs offset c * LD + tap
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) {
        const int kk = ct * 32 + l31;
        offb[ct] = (kk / 3) * LD + (kk % 3);
    }
    float accb = 0.0f;
    const int nq = (T + 7) / 8;
    for (int b = 0; b < B; ++b) {
        __syncthreads();
        stage_h(hs, h, b, T, LD, tid);
        __syncthreads();
        const float *dr = dout + ((int64_t)b * M + p0 + l31) * T;
        for (int q = 0; q < nq; ++q) {
            const int t0 = 8 * q + 4 * hi;
            float4 dv;
            if (ALIGNED && t0 + 3 < T) dv = *reinterpret_cast<const float4 *>(dr + t0);
            else dv = make_float4(t0 < T ? dr[t0] : 0.0f, t0 + 1 < T ? dr[t0 + 1] : 0.0f, t0 + 2 < T ? dr[t0 + 2] : 0.0f, t0 + 3 < T ? dr[t0 + 3] : 0.0f);
            accb += (dv.x + dv.y) + (dv.z + dv.w);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tt = min(t0 + j, T - 1);                      // beyond T the A value is 0: any finite B value will do
                const float av = f4c(dv, j);
#pragma unroll
                for (int ct = 0; ct < 6; ++ct) acc[ct] = mfma32(av, hs[offb[ct] + tt], acc[ct]);      // h[c, tt + tap - 1]
            }
        }
    }
    if (dW) {
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) dW[(int64_t)(p0 + drow(r, hi)) * KK + ct * 32 + l31] = acc[ct][r];
    }
    accb += __shfl_xor(accb, 32, 64);
    if (dbias && hi == 0) dbias[p0 + l31] = accb;
}

// ---- dh, first pass: workgroup = (slice of the rows p, utterance b): G_part[(c,k), t] = sum over the slice of W[p,(c,k)] dout[b,p,t];
//      wave = one 32-column tile of t (T <= 128), all six 32-row tiles of (c,k); W and dout go through LDS 32 rows at a time ----------
__global__ void __launch_bounds__(256, 2) k_kc_dh(const float *__restrict__ W, const float *__restrict__ dout, float *__restrict__ part,
                                                  int B, int M, int T, int LDD, int prows)
{
    extern __shared__ float sm[];
    float *ws = sm, *dsm = sm + 32 * KK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int ks = blockIdx.x, b = blockIdx.y, pbeg = ks * prows;
    const int tcol = wave * 32 + l31, tc = min(tcol, T - 1);
    const bool wave_live = wave * 32 < T;
    f32x16 acc[6];
#pragma unroll
    for (int rt = 0; rt < 6; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
    for (int pc = pbeg; pc < pbeg + prows; pc += 32) {
        __syncthreads();
        for (int idx = tid; idx < 32 * (KK / 4); idx += 256)
            reinterpret_cast<float4 *>(ws)[idx] = reinterpret_cast<const float4 *>(W + (int64_t)pc * KK)[idx];
        for (int idx = tid; idx < 32 * T; idx += 256) {
            const int row = idx / T, t = idx - row * T;
            dsm[row * LDD + t] = dout[((int64_t)b * M + pc + row) * T + t];
        }
        __syncthreads();
        if (wave_live) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int kp = 2 * s + hi;                               // row p of the chunk = the reduction index
                const float bv = dsm[kp * LDD + tc];
#pragma unroll
                for (int rt = 0; rt < 6; ++rt) acc[rt] = mfma32(ws[kp * KK + rt * 32 + l31], bv, acc[rt]);
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
__global__ void __launch_bounds__(256) k_kc_dh_fold(const float *__restrict__ part, float *__restrict__ dh, int B, int T, int nks)
{
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
    dh[idx] = v;
}

}  // namespace fdk_kconv

namespace fdk {
using namespace fdk_kconv;

// LDS row stride of the h window: >= T + 2 and == 3 (mod 32), so that the dW kernel's B reads -- lane = column (c, tap), address
// c * LD + tap + t -- fall into 32 different banks (c * 3 + tap = the lane's column index)
static int kconv_ld(int T) { return ((T + 2 + 28) / 32) * 32 + 3; }
bool kconv_supported(int M, int T) { return M > 0 && M % 128 == 0 && T >= 1 && T <= 128; }
size_t kconv_scratch_floats(int B, int T) { return (size_t)16 * B * KK * T; }      // the 16 row slices of the dh pass

hipError_t kconv_forward(const Launch &L, const float *h, const float *W, const float *bias, float *out, int B, int M, int T)
{
    const int LD = kconv_ld(T), gx = M / 128;
    int ny = std::max(1, std::min(B, (2 * L.ctx->num_cus + gx - 1) / gx));
    const int bchunk = (B + ny - 1) / ny;
    ny = (B + bchunk - 1) / bchunk;
    FD_LAUNCH(L, "kconv_forward", k_kc_fwd, dim3(gx, ny), dim3(256), sizeof(float) * CI * LD, h, W, bias, out, B, M, T, LD, bchunk);
    return hipSuccess;
}

hipError_t kconv_backward(const Launch &L, const float *h, const float *W, const float *dout, float *dh, float *dW, float *dbias, int B, int M,
                          int T, float *scratch)
{
    const int LD = kconv_ld(T);
    if (dW || dbias) {
        if (T % 4 == 0) FD_LAUNCH(L, "kconv_backward_w", k_kc_dw<true>, dim3(M / 128), dim3(256), sizeof(float) * CI * LD, h, dout, dW, dbias, B, M, T, LD);
        else FD_LAUNCH(L, "kconv_backward_w", k_kc_dw<false>, dim3(M / 128), dim3(256), sizeof(float) * CI * LD, h, dout, dW, dbias, B, M, T, LD);
    }
    if (dh) {
        int nks = 16;
        while (nks > 1 && (M % (nks * 32) != 0)) nks >>= 1;              // slices of whole 32-row chunks
        const int LDD = T | 1;                                            // odd row stride: the staging writes spread over the banks
        FD_LAUNCH(L, "kconv_backward_h", k_kc_dh, dim3(nks, B), dim3(256), sizeof(float) * (32 * KK + 32 * LDD), W, dout, scratch, B, M, T, LDD, M / nks);
        FD_LAUNCH(L, "kconv_backward_h_fold", k_kc_dh_fold, dim3((B * CI * T + 255) / 256), dim3(256), 0, (const float *)scratch, dh, B, T, nks);
    }
    return hipSuccess;
}

}  // namespace fdk
