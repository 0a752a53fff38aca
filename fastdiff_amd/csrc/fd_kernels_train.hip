// fd_kernels_train.hip -- the location-variable convolution as a differentiable OPERATOR: forward and the three
// gradients of   out[b,o,q] = bias[b,o,l] + sum_{i,k} xpad[b,i,q + k - pad] * K[b,i,o,k,l],   l = q / hop,   pad = (ks - 1) / 2
// (modules/FastDiff/module/modules.py:220-253 with dilation = 1, the only value the model passes, modules.py:216), in the REFERENCE's
// own tensor layouts: x [B,Cin,L], K [B,Cin,Cout,ks,T] (T innermost), bias [B,Cout,T], out [B,Cout,L], L = T * hop.  This is what the
// reference's training step (theta_timestep_loss, util.py:291-325) differentiates through twelve times per forward; everything else
// of that step can stay on PyTorch autograd around it.  Plain fp32 VALU kernels: correctness first, the training side is not the
// measured hot path.
#include "fd_kernels.h"

namespace fdk_train {

// thread = one output (b, o, q); lanes run along q: x reads are coalesced, the K element is the same for every lane of a frame
__global__ void __launch_bounds__(256) k_lvc_fwd(const float *__restrict__ x, const float *__restrict__ K, const float *__restrict__ bias,
                                                 float *__restrict__ out, int Cin, int Cout, int ks, int T, int hop)
{
    const int L = T * hop, pad = (ks - 1) / 2;
    const int q = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (q >= L) return;
    const int l = q / hop;
    float acc = bias[((int64_t)b * Cout + o) * T + l];
    for (int i = 0; i < Cin; ++i) {
        const float *xr = x + ((int64_t)b * Cin + i) * L;
        const float *kr = K + ((((int64_t)b * Cin + i) * Cout + o) * ks) * T + l;
        for (int k = 0; k < ks; ++k) {
            const int p = q + k - pad;
            if (p >= 0 && p < L) acc = fmaf(xr[p], kr[(int64_t)k * T], acc);
        }
    }
    out[((int64_t)b * Cout + o) * L + q] = acc;
}

// dx[b,i,p] = sum_{o,k} dout[b,o,q] * K[b,i,o,k,q/hop],  q = p - k + pad
__global__ void __launch_bounds__(256) k_lvc_bwd_x(const float *__restrict__ dout, const float *__restrict__ K, float *__restrict__ dx,
                                                   int Cin, int Cout, int ks, int T, int hop)
{
    const int L = T * hop, pad = (ks - 1) / 2;
    const int p = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (p >= L) return;
    float acc = 0.0f;
    for (int k = 0; k < ks; ++k) {
        const int q = p - k + pad;
        if (q < 0 || q >= L) continue;
        const int l = q / hop;
        for (int o = 0; o < Cout; ++o)
            acc = fmaf(dout[((int64_t)b * Cout + o) * L + q], K[((((int64_t)b * Cin + i) * Cout + o) * ks + k) * T + l], acc);
    }
    dx[((int64_t)b * Cin + i) * L + p] = acc;
}

// dK[b,i,o,k,l] = sum_{s<hop} dout[b,o,l*hop+s] * xpad[b,i,l*hop+s+k-pad];   dbias[b,o,l] = sum_s dout[b,o,l*hop+s].
// Workgroup = one frame (b, l): the frame's columns go through LDS 64 at a time; thread t owns the outputs e = t, t + 256, ... of the
// Cin*Cout*ks products (e = (i*Cout + o)*ks + k) and, for t < Cout, the bias gradient of row t.
constexpr int DK_CHUNK = 64, DK_MAXE = 32;      // up to 256 * 32 = 8192 kernel elements per frame (the model: 32*64*3 = 6144)
__global__ void __launch_bounds__(256) k_lvc_bwd_k(const float *__restrict__ x, const float *__restrict__ dout, float *__restrict__ dK,
                                                   float *__restrict__ dbias, int Cin, int Cout, int ks, int T, int hop)
{
    extern __shared__ float sm[];      // dout tile [Cout][DK_CHUNK], x tile [Cin][DK_CHUNK + ks - 1]
    const int L = T * hop, pad = (ks - 1) / 2, XW = DK_CHUNK + ks - 1;
    float *sd = sm, *sx = sm + Cout * DK_CHUNK;
    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int nE = Cin * Cout * ks;
    float acc[DK_MAXE], accb = 0.0f;
#pragma unroll
    for (int j = 0; j < DK_MAXE; ++j) acc[j] = 0.0f;
    for (int s0 = 0; s0 < hop; s0 += DK_CHUNK) {
        const int n = min(DK_CHUNK, hop - s0), q0 = l * hop + s0;
        __syncthreads();
        for (int e = tid; e < Cout * DK_CHUNK; e += 256) {
            const int o = e / DK_CHUNK, s = e - o * DK_CHUNK;
            sd[e] = s < n ? dout[((int64_t)b * Cout + o) * L + q0 + s] : 0.0f;
        }
        for (int e = tid; e < Cin * XW; e += 256) {
            const int i = e / XW, s = e - i * XW, p = q0 + s - pad;
            sx[e] = (s < n + ks - 1 && p >= 0 && p < L) ? x[((int64_t)b * Cin + i) * L + p] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DK_MAXE; ++j) {
            const int e = tid + j * 256;
            if (e < nE) {
                const int k = e % ks, io = e / ks, o = io % Cout, i = io / Cout;
                const float *dr = sd + o * DK_CHUNK, *xr = sx + i * XW + k;
                float a = acc[j];
                for (int s = 0; s < DK_CHUNK; ++s) a = fmaf(dr[s], xr[s], a);
                acc[j] = a;
            }
        }
        if (tid < Cout)
            for (int s = 0; s < DK_CHUNK; ++s) accb += sd[tid * DK_CHUNK + s];
    }
    if (dK) {
#pragma unroll
        for (int j = 0; j < DK_MAXE; ++j) {
            const int e = tid + j * 256;
            if (e < nE) dK[((int64_t)b * nE + e) * T + l] = acc[j];
        }
    }
    if (dbias && tid < Cout) dbias[((int64_t)b * Cout + tid) * T + l] = accb;
}

}  // namespace fdk_train

namespace fdk {
using namespace fdk_train;

hipError_t lvc_op_forward(const Launch &L, const float *x, const float *K, const float *bias, float *out, int B, int Cin, int Cout, int ks,
                          int T, int hop)
{
    const int Ln = T * hop;
    FD_LAUNCH(L, "lvc_op_forward", k_lvc_fwd, dim3((Ln + 255) / 256, Cout, B), dim3(256), 0, x, K, bias, out, Cin, Cout, ks, T, hop);
    return hipSuccess;
}

hipError_t lvc_op_backward(const Launch &L, const float *x, const float *K, const float *dout, float *dx, float *dK, float *dbias, int B,
                           int Cin, int Cout, int ks, int T, int hop)
{
    const int Ln = T * hop;
    if (dx) FD_LAUNCH(L, "lvc_op_backward_x", k_lvc_bwd_x, dim3((Ln + 255) / 256, Cin, B), dim3(256), 0, dout, K, dx, Cin, Cout, ks, T, hop);
    if (dK || dbias) {
        const size_t shmem = sizeof(float) * ((size_t)Cout * DK_CHUNK + (size_t)Cin * (DK_CHUNK + ks - 1));
        FD_LAUNCH(L, "lvc_op_backward_k", k_lvc_bwd_k, dim3(T, B), dim3(256), shmem, x, dout, dK, dbias, Cin, Cout, ks, T, hop);
    }
    return hipSuccess;
}

}  // namespace fdk
