// fd_kernels_convt.hip -- a6 the LVC block's ConvTranspose1d up-sampler (modules.py:163-166,205-206)
// (one stage of the gfx950 kernel set; shared device helpers: fd_kernels_common.h; the one-thread-per-output twins: fd_kernels_naive.hip)
#include "fd_kernels_common.h"

namespace fdk_fast {

// =================================================================================================
// a6: ConvTranspose1d(32,32,2r,stride r,pad r/2) of leaky_relu(x,0.2) (modules.py:163-166,205-206)
// =================================================================================================
// out[o, q*R + ph] = b[o] + sum_i x[i, q + offA]*W[i, o, kA] + x[i, q + offB]*W[i, o, kB]: for each of the R output phases a
// 32x64 by 64x(columns) product on the matrix pipe, rows = output channel, cols = input position q, k = (tap select, i).
// A operands (per-phase weight slices) are pre-packed [phase][s4][lane][4]; B comes from an LDS window of leaky_relu(x).
constexpr int CT_LD = 132;     // 128 input positions + 1 halo each side, padded

template <int R>
__global__ void __launch_bounds__(256, 2) k_convt(const float *__restrict__ xin, const float *__restrict__ pack,
                                               const float *__restrict__ bias, float *__restrict__ out, int Lin,
                                               const int *__restrict__ run_if, const int *__restrict__ lens, int per_frame)
{
    __shared__ float xs[fd::C * CT_LD];
    if (run_if && *run_if == 0) return;      // fallback launch behind k_convt_h2
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q0 = blockIdx.x * 128, Lout = Lin * R;
    const int Lb = lens ? lens[b] * per_frame : Lin;      // this utterance's own input length
    if (q0 >= Lb) return;
    {
        constexpr int TOTAL = fd::C * 130, NK = (TOTAL + 255) / 256;
        float v[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid, ci = idx / 130, jj = idx - ci * 130, j = q0 - 1 + jj;
            v[k] = (idx < TOTAL && j >= 0 && j < Lb) ? lrelu(xin[((int64_t)b * fd::C + ci) * Lin + j], 0.2f) : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid, ci = idx / 130, jj = idx - ci * 130;
            if (idx < TOTAL) xs[ci * CT_LD + jj] = v[k];
        }
    }
    __syncthreads();
    const int ql = wave * 32 + l31, q = q0 + ql;
    if (q0 + wave * 32 >= Lb) return;
    float4 cb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cb[j] = reinterpret_cast<const float4 *>(bias)[2 * j + hi];
    float *ob = out + ((int64_t)b * fd::C + 4 * hi) * Lout + (int64_t)q * R;
    const unsigned Lu = (unsigned)Lout;
    float4 wa[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wa[0][i] = reinterpret_cast<const float4 *>(pack)[i * 64 + lane];
    f32x16 acc[R];
#pragma unroll
    for (int ph = 0; ph < R; ++ph) {
        if (ph + 1 < R) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wa[(ph + 1) & 1][i] = reinterpret_cast<const float4 *>(pack)[((ph + 1) * 8 + i) * 64 + lane];
        }
        const int offA = (ph < R / 2) ? 0 : 1, offB = offA - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ph][r] = f4c(cb[r >> 2], r & 3);
#pragma unroll
        for (int s = 0; s < 32; ++s) {          // kk = 2s+hi = sel*32 + i
            const int sel = s >> 4, i = ((2 * s) & 31) + hi;
            acc[ph] = mfma32(f4c(wa[ph & 1][s >> 2], s & 3), xs[i * CT_LD + 1 + ql + (sel ? offB : offA)], acc[ph]);
        }
    }
    // a lane holds the R consecutive outputs q*R .. q*R+R-1 of 16 channels: 16 B stores, 32 lanes cover 32*R contiguous floats
    if (q < Lb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float *dst = ob + (unsigned)((r & 3) + 8 * (r >> 2)) * Lu;
#pragma unroll
            for (int p4 = 0; p4 < R; p4 += 4)
                *reinterpret_cast<float4 *>(dst + p4) = make_float4(acc[p4][r], acc[p4 + 1][r], acc[p4 + 2][r], acc[p4 + 3][r]);
        }
    }
}

// The same ConvTranspose on the fp16 matrix pipe with 2-piece operands: 12 MFMAs of 32 cycles per
// output phase instead of 32 of 64.  leaky_relu(x) is split once into a [position][piece][32 ch] fp16 image (row = q - q0 + 1).
#ifndef FD_CONVT_OCC
#define FD_CONVT_OCC(R) 2      // workgroups per CU the register budget is cut for.  r = 4 fits three (168 VGPRs, no spills) and is
                               // slower with them: 73 -> 80 us in the step (profiles/r03/s44_convt_occupancy.txt)
#endif
template <int R>
__global__ void __launch_bounds__(256, FD_CONVT_OCC(R)) k_convt_h2(const float *__restrict__ xin, const float4 *__restrict__ pack16,
                                                  const float *__restrict__ bias, float *__restrict__ out, int Lin,
                                                  int *__restrict__ range_flag, const int *__restrict__ lens, int per_frame)
{
    __shared__ __attribute__((aligned(16))) char xs[130 * 128];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q0 = blockIdx.x * 128, Lout = Lin * R;
    const int Lb = lens ? lens[b] * per_frame : Lin;      // this utterance's own input length
    if (q0 >= Lb || skip_after_previous_overflow(range_flag)) return;
    float mx = 0.0f;
    {   // thread = (8-channel group, position): 130 positions x 4 groups = 520 units
        float v[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = k * 256 + tid, cg = u / 130, jj = u - cg * 130, j = q0 - 1 + jj;
            const bool ok = u < 520 && j >= 0 && j < Lb;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[k][c] = ok ? xin[((int64_t)b * fd::C + cg * 8 + c) * Lin + j] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = k * 256 + tid, cg = u / 130, jj = u - cg * 130;
            if (u < 520) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { mx = fmaxf(mx, fabsf(v[k][c])); v[k][c] = lrelu(v[k][c], 0.2f); }
                float4 ph, pl;
                split8(v[k], ph, pl);
                *reinterpret_cast<float4 *>(xs + h2_off(jj, cg)) = ph;
                *reinterpret_cast<float4 *>(xs + h2_off(jj, 4 + cg)) = pl;
            }
        }
    }
    if (!(mx < GX_LIMIT)) atomicOr(range_flag, 1);
    __syncthreads();
    const int ql = wave * 32 + l31, q = q0 + ql;
    if (q0 + wave * 32 >= Lb) return;
    float4 cb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cb[j] = reinterpret_cast<const float4 *>(bias)[2 * j + hi];
    float *ob = out + ((int64_t)b * fd::C + 4 * hi) * Lout + (int64_t)q * R;
    const unsigned Lu = (unsigned)Lout;
    // B operands: rows ql (position q-1), ql+1 (q), ql+2 (q+1); two channel halves, two pieces each
    float4 bx[3][2][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int p = 0; p < 2; ++p) bx[d][c2][p] = *reinterpret_cast<const float4 *>(xs + h2_off(ql + d, p * 4 + c2 * 2 + hi));
    float4 wa[2][2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) wa[0][p][kg] = pack16[(p * 4 + kg) * 64 + lane];
    // a lane ends up with the R consecutive outputs q*R .. q*R+R-1 of 16 channels: phases are done four at a time and
    // leave as 16 B stores (32 lanes cover 32*R contiguous floats per channel)
#pragma unroll
    for (int pg = 0; pg < R; pg += 4) {
        float res[4][16];
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int ph = pg + pi;
            if (ph + 1 < R) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int kg = 0; kg < 4; ++kg) wa[(ph + 1) & 1][p][kg] = pack16[(((ph + 1) * 2 + p) * 4 + kg) * 64 + lane];
            }
            const int offA = (ph < R / 2) ? 0 : 1, offB = offA - 1;      // sel 0 reads position q + offA, sel 1 position q + offB
            f32x16 ah, al;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ah[r] = f4c(cb[r >> 2], r & 3); al[r] = 0.0f; }
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {          // k = 16*kg + 8*hi + e = sel*32 + i
                const int d = 1 + ((kg >> 1) ? offB : offA), c2 = kg & 1;
                ah = mfma_f16(wa[ph & 1][0][kg], bx[d][c2][0], ah);
                al = mfma_f16(wa[ph & 1][0][kg], bx[d][c2][1], al);
                al = mfma_f16(wa[ph & 1][1][kg], bx[d][c2][0], al);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) res[pi][r] = fmaf(al[r], GX_INV_SCALE, ah[r]);
        }
        if (q < Lb) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                lvc_st<32>(reinterpret_cast<float4 *>(ob + (unsigned)((r & 3) + 8 * (r >> 2)) * Lu + pg), make_float4(res[0][r], res[1][r], res[2][r], res[3][r]));
        }
    }
}

}  // namespace fdk_fast

// ------------------------------------------------------------------------------------------------
// stage drivers
// ------------------------------------------------------------------------------------------------
namespace fdk {
using namespace fdk_fast;

hipError_t fast_convt(const Launch &L, int n, const float *x_in, float *x_out, int B, int Lin)
{
    const DevWeights &w = L.ctx->w;
    const dim3 grid((Lin + 127) / 128, B);
    fd_context *c = L.ctx;
    const int *run_if = nullptr;
    const char *n8 = "convt_r8", *n4 = "convt_r4";
    const Pipe pipe = fd_pipe(c, c->conv_f16 && w.convt_f16_ok, 16 + n);
    if (pipe != PIPE_F32_ONLY) {
        int *flag = c->ws.range_flag + 16 + n;
        if (fd::ratio(n) == 8)
            FD_LAUNCH(L, n8, k_convt_h2<8>, grid, dim3(256), 0, x_in, reinterpret_cast<const float4 *>(w.up_h2[n]), w.blk[n].up.b, x_out, Lin, flag, c->step_lens, fd::hop(n) / fd::ratio(n));
        else
            FD_LAUNCH(L, n4, k_convt_h2<4>, grid, dim3(256), 0, x_in, reinterpret_cast<const float4 *>(w.up_h2[n]), w.blk[n].up.b, x_out, Lin, flag, c->step_lens, fd::hop(n) / fd::ratio(n));
        run_if = flag;
        n8 = n4 = "convt_fp32_fallback";
        if (pipe == PIPE_F16_ONLY) return hipSuccess;
    }
    if (fd::ratio(n) == 8)
        FD_LAUNCH(L, n8, k_convt<8>, grid, dim3(256), 0, x_in, w.up_pack[n], w.blk[n].up.b, x_out, Lin, run_if, c->step_lens, fd::hop(n) / fd::ratio(n));
    else
        FD_LAUNCH(L, n4, k_convt<4>, grid, dim3(256), 0, x_in, w.up_pack[n], w.blk[n].up.b, x_out, Lin, run_if, c->step_lens, fd::hop(n) / fd::ratio(n));
    return hipSuccess;
}

}  // namespace fdk
