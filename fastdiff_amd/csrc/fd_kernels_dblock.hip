// fd_kernels_dblock.hip -- a4 DiffusionDBlock (modules.py:116-138)
// (one stage of the gfx950 kernel set; shared device helpers: fd_kernels_common.h; the one-thread-per-output twins: fd_kernels_naive.hip)
#include "fd_kernels_common.h"

namespace fdk_fast {

// =================================================================================================
// a4: DiffusionDBlock (modules.py:127-138), fused: strided pick, 3 dilated convs, 1x1 residual, add
// One workgroup = one 128-column tile at the DOWN-sampled rate; every layer is computed on all 128 columns
// and the valid region shrinks by the dilation (1+2+4 = 7 per side), so tiles advance by 114 columns.
// =================================================================================================
constexpr int DB_LD = 136;   // 128 + 4 guard columns each side (max dilation 4)
constexpr int DB_STRIDE = 114;

template <int DIL, bool LRELU>
__device__ __forceinline__ void conv96_tile(f32x16 &acc, const float4 (&wa)[12], const float *in, int ld, int col, int hi)
{
    // 48 k-steps: kk = 2s+hi = tap*32 + ci
    const int o[3] = {opaque(hi * ld + col - DIL), opaque(hi * ld + col), opaque(hi * ld + col + DIL)};
#pragma unroll
    for (int s = 0; s < 48; ++s) {
        const int tap = s >> 4, c2 = (2 * s) & 31;
        float v = in[o[tap] + c2 * ld];
        if (LRELU) v = lrelu(v, 0.2f);
        acc = mfma32(f4c(wa[s >> 2], s & 3), v, acc);
    }
}

template <int F>
__global__ void __launch_bounds__(256, 2) k_dblock(const float *__restrict__ xin, float *__restrict__ out,
                                                   const float *__restrict__ p0, const float *__restrict__ p1,
                                                   const float *__restrict__ p2, const float *__restrict__ pr,
                                                   const float *__restrict__ b0, const float *__restrict__ b1,
                                                   const float *__restrict__ b2, const float *__restrict__ br, int Lin, int Lo,
                                                   const int *__restrict__ run_if, const int *__restrict__ lens, int per_frame)
{
    __shared__ __attribute__((aligned(16))) float xs[fd::C * DB_LD];
    if (run_if && *run_if == 0) return;      // fallback launch behind k_dblock_h2: only when that kernel flagged its operands
    __shared__ __attribute__((aligned(16))) float hA[fd::C * DB_LD];
    __shared__ __attribute__((aligned(16))) float hB[fd::C * DB_LD];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int pbase = blockIdx.x * DB_STRIDE - 7;   // down-sampled position of tile column 0
    const int Lob = lens ? lens[b] * per_frame : Lo;      // this utterance's own length at the output rate
    if (blockIdx.x * DB_STRIDE >= Lob) return;
    // stage the strided pick x[..., ::F]; zero outside [0, Lo) and in the guard columns (loads batched ahead of the writes)
    {
        constexpr int NK = fd::C * DB_LD / 256;     // 17
        float v[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid, ci = idx / DB_LD, cc = idx - ci * DB_LD, p = pbase + cc - 4;
            v[k] = (cc >= 4 && cc < 132 && p >= 0 && p < Lob) ? xin[((int64_t)b * fd::C + ci) * Lin + (int64_t)p * F] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid;
            xs[idx] = v[k];
            hA[idx] = 0.0f;
            hB[idx] = 0.0f;
        }
    }
    __syncthreads();
    const int c = wave * 32 + l31;        // this lane's tile column
    const int p = pbase + c;              // its down-sampled position
    const bool inside = (p >= 0 && p < Lob);
    float4 wa[12];
    f32x16 acc;
    // layer 1: dil 1 on leaky_relu(xs)
#pragma unroll
    for (int i = 0; i < 12; ++i) wa[i] = reinterpret_cast<const float4 *>(p0)[i * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b0[drow(r, hi)];
    conv96_tile<1, true>(acc, wa, xs, DB_LD, 4 + c, hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) hA[drow(r, hi) * DB_LD + 4 + c] = inside ? acc[r] : 0.0f;
    __syncthreads();
    // layer 2: dil 2
#pragma unroll
    for (int i = 0; i < 12; ++i) wa[i] = reinterpret_cast<const float4 *>(p1)[i * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b1[drow(r, hi)];
    conv96_tile<2, true>(acc, wa, hA, DB_LD, 4 + c, hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) hB[drow(r, hi) * DB_LD + 4 + c] = inside ? acc[r] : 0.0f;
    __syncthreads();
    // layer 3: dil 4, plus the 1x1 residual on the raw pick (residual_dense commutes with the nearest pick)
#pragma unroll
    for (int i = 0; i < 12; ++i) wa[i] = reinterpret_cast<const float4 *>(p2)[i * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b2[drow(r, hi)] + br[drow(r, hi)];
    conv96_tile<4, true>(acc, wa, hB, DB_LD, 4 + c, hi);
    {
        float4 wr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wr[i] = reinterpret_cast<const float4 *>(pr)[i * 64 + lane];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = mfma32(f4c(wr[s >> 2], s & 3), xs[(2 * s + hi) * DB_LD + 4 + c], acc);
    }
    if (inside && c >= 7 && c < 7 + DB_STRIDE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((int64_t)b * fd::C + drow(r, hi)) * Lo + p] = acc[r];
    }
}

// The same DBlock on the fp16 matrix pipe with 2-piece operands: 57 MFMAs of 32 cycles per wave
// instead of 160 of 64.  Activations live in LDS as [column][piece][32 ch] fp16 images (128 B per column, slots swizzled as
// in k_lvc_h2); an image holds leaky_relu of the layer output because that is the only form the next layer reads; the raw
// pick of x gets its own image for the 1x1 residual.  Same tiling: 128 columns, 114 valid.
constexpr int DBH_ROWS = 136;        // 128 columns + 4 zero guard columns each side (row = column + 4)

template <int DIL>
__device__ __forceinline__ void conv96_h2(f32x16 &ah, f32x16 &al, const float4 (&wa)[2][6], const char *img, int c, int hi)
{
    int off[3][2][2];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) off[tap][p][c2] = h2_off(4 + c + (tap - 1) * DIL, p * 4 + c2 * 2 + hi);
#pragma unroll
    for (int kg = 0; kg < 6; ++kg) {
        const float4 b1 = *reinterpret_cast<const float4 *>(img + off[kg >> 1][0][kg & 1]);
        const float4 b2 = *reinterpret_cast<const float4 *>(img + off[kg >> 1][1][kg & 1]);
        ah = mfma_f16(wa[0][kg], b1, ah);
        al = mfma_f16(wa[0][kg], b2, al);
        al = mfma_f16(wa[1][kg], b1, al);
    }
}
// write leaky_relu(hi + 2^-11 lo) of a 32x32 tile (0 where `inside` is false) as the two pieces of column c
__device__ __forceinline__ void store_act_h2(char *img, const f32x16 &ah, const f32x16 &al, int c, int hi, bool inside, float &mx)
{
    const int row = 4 + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                         // D rows 8j + 4hi + {0..3}: half a slot
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = inside ? lrelu(fmaf(al[4 * j + i], GX_INV_SCALE, ah[4 * j + i]), 0.2f) : 0.0f;
            mx = fmaxf(mx, fabsf(v[i]));
        }
        uint2 ph, pl;
        split2(v[0], v[1], ph.x, pl.x);
        split2(v[2], v[3], ph.y, pl.y);
        *reinterpret_cast<uint2 *>(img + h2_off(row, j) + 8 * hi) = ph;
        *reinterpret_cast<uint2 *>(img + h2_off(row, 4 + j) + 8 * hi) = pl;
    }
}

// AUDIO (first DBlock only): its input is first_audio_conv(x), a 1 -> 32 channel k7 conv of which it uses every F-th column.  Those
// columns are recomputed from the audio (7 samples, 56 FMAs per 8 channels; same operation order as k_first_conv, so the same
// bits) instead of picked out of the 32-channel tensor: a stride-F pick of fp32 fetches every cache line of it, 226 MB at the
// benchmark size against 7 MB of audio.
#ifndef FD_DBLOCK_OCC
#define FD_DBLOCK_OCC 3      // waves per SIMD the register allocation is held to: three workgroups per CU fit the 52 KB of LDS, and the kernel is
#endif                       // a chain of three dependent layers with barriers between them (A/B in one session: 73.2 -> 69.9 us at B=8, 18.2 -> 15.4 at B=1)
template <int F, bool AUDIO>
__global__ void __launch_bounds__(256, FD_DBLOCK_OCC) k_dblock_h2(const float *__restrict__ xin, float *__restrict__ out,
                                                      const float4 *__restrict__ p0, const float4 *__restrict__ p1,
                                                      const float4 *__restrict__ p2, const float4 *__restrict__ pr,
                                                      const float *__restrict__ b0, const float *__restrict__ b1,
                                                      const float *__restrict__ b2, const float *__restrict__ br, int Lin, int Lo,
                                                      int *__restrict__ range_flag, const int *__restrict__ lens, int per_frame,
                                                      const float *__restrict__ audio, const float *__restrict__ fw,
                                                      const float *__restrict__ fb)
{
    __shared__ __attribute__((aligned(16))) char xl[DBH_ROWS * 128];      // leaky_relu(x pick); later the layer-2 output
    __shared__ __attribute__((aligned(16))) char xr[DBH_ROWS * 128];      // raw x pick (1x1 residual)
    __shared__ __attribute__((aligned(16))) char ha[DBH_ROWS * 128];      // layer-1 output
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int pbase = blockIdx.x * DB_STRIDE - 7;   // down-sampled position of tile column 0
    const int Lob = lens ? lens[b] * per_frame : Lo;      // this utterance's own length at the output rate
    if (blockIdx.x * DB_STRIDE >= Lob || skip_after_previous_overflow(range_flag)) return;
    float mx = 0.0f;
    // AUDIO: the first conv's weights (224) and biases (32) through vector loads and LDS, not through scalar loads of a uniform index:
    // scalar DATA loads are what a short-lived neighbour process on the same compute units can disturb (k_first_conv, LABBOOK.md section 4)
    __shared__ float fwl[AUDIO ? 256 : 1];
    if constexpr (AUDIO) {
        fwl[tid] = tid < 224 ? fw[tid] : fb[tid - 224];
        __syncthreads();
    }
    // ---- stage the strided pick x[..., ::F]: thread = (8-channel group, column), two columns per thread; zero outside [0, Lo)
    {
        float v[2][8];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = k * 256 + tid, cg = u >> 7, cc = u & 127, p = pbase + cc;
            const bool ok = p >= 0 && p < Lob;
            if (AUDIO) {
                const int Lb = Lob * F, t = p * F;                // the utterance's audio length; this column's sample
                const float *xa = audio + (int64_t)b * Lin;
                float xv[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) xv[i] = (ok && t - 3 + i >= 0 && t - 3 + i < Lb) ? xa[t - 3 + i] : 0.0f;
                const int o0 = __builtin_amdgcn_readfirstlane(cg) * 8;     // the channel group is uniform over a wave
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float r = fwl[224 + o0 + c];
#pragma unroll
                    for (int i = 0; i < 7; ++i) r += fwl[(o0 + c) * 7 + i] * xv[i];
                    v[k][c] = ok ? r : 0.0f;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[k][c] = ok ? xin[((int64_t)b * fd::C + cg * 8 + c) * Lin + (int64_t)p * F] : 0.0f;
            }
        }
        if (tid < 128) {        // the 8 guard columns of all three images: zeros
            const int g = tid >> 4, row = g < 4 ? g : 128 + g, part = tid & 15;      // 16 x 8 B per 128 B row
            *reinterpret_cast<uint2 *>(xl + row * 128 + part * 8) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2 *>(xr + row * 128 + part * 8) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2 *>(ha + row * 128 + part * 8) = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = k * 256 + tid, cg = u >> 7, cc = u & 127;
            float a[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) { mx = fmaxf(mx, fabsf(v[k][c])); a[c] = lrelu(v[k][c], 0.2f); }
            float4 ph, pl;
            split8(v[k], ph, pl);
            *reinterpret_cast<float4 *>(xr + h2_off(4 + cc, cg)) = ph;
            *reinterpret_cast<float4 *>(xr + h2_off(4 + cc, 4 + cg)) = pl;
            split8(a, ph, pl);
            *reinterpret_cast<float4 *>(xl + h2_off(4 + cc, cg)) = ph;
            *reinterpret_cast<float4 *>(xl + h2_off(4 + cc, 4 + cg)) = pl;
        }
    }
    __syncthreads();
    const int c = wave * 32 + l31;        // this lane's tile column
    const int p = pbase + c;              // its down-sampled position
    const bool inside = (p >= 0 && p < Lob);
    float4 wa[2][6];
    f32x16 ah, al;
    auto load_w = [&](const float4 *pk) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int kg = 0; kg < 6; ++kg) wa[q][kg] = pk[(q * 6 + kg) * 64 + lane];
    };
    // layer 1: dil 1 on leaky_relu(x)
    load_w(p0);
#pragma unroll
    for (int r = 0; r < 16; ++r) { ah[r] = b0[drow(r, hi)]; al[r] = 0.0f; }
    conv96_h2<1>(ah, al, wa, xl, c, hi);
    store_act_h2(ha, ah, al, c, hi, inside, mx);
    __syncthreads();
    // layer 2: dil 2
    load_w(p1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { ah[r] = b1[drow(r, hi)]; al[r] = 0.0f; }
    conv96_h2<2>(ah, al, wa, ha, c, hi);
    store_act_h2(xl, ah, al, c, hi, inside, mx);      // xl is free: layer 1 was its only reader
    __syncthreads();
    // layer 3: dil 4, plus the 1x1 residual on the raw pick (residual_dense commutes with the nearest pick)
    load_w(p2);
#pragma unroll
    for (int r = 0; r < 16; ++r) { ah[r] = b2[drow(r, hi)] + br[drow(r, hi)]; al[r] = 0.0f; }
    conv96_h2<4>(ah, al, wa, xl, c, hi);
    {
        float4 wr[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) wr[q][kg] = pr[(q * 2 + kg) * 64 + lane];
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {      // k = input channel: 16*kg + 8*hi + e
            const float4 x1 = *reinterpret_cast<const float4 *>(xr + h2_off(4 + c, kg * 2 + hi));
            const float4 x2 = *reinterpret_cast<const float4 *>(xr + h2_off(4 + c, 4 + kg * 2 + hi));
            ah = mfma_f16(wr[0][kg], x1, ah);
            al = mfma_f16(wr[0][kg], x2, al);
            al = mfma_f16(wr[1][kg], x1, al);
        }
    }
    if (inside && c >= 7 && c < 7 + DB_STRIDE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((int64_t)b * fd::C + drow(r, hi)) * Lo + p] = fmaf(al[r], GX_INV_SCALE, ah[r]);
    }
    if (!(mx < GX_LIMIT)) atomicOr(range_flag, 1);
}

}  // namespace fdk_fast

// ------------------------------------------------------------------------------------------------
// stage drivers
// ------------------------------------------------------------------------------------------------
namespace fdk {
using namespace fdk_fast;

hipError_t fast_dblock(const Launch &L, int d, int B, int T, const float *audio)
{
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    int Lin = T * fd::HOPT;
    for (int i = 0; i < d; ++i) Lin /= fd::down_factor(i);
    const int f = fd::down_factor(d), Lo = Lin / f;
    const dim3 grid((Lo + DB_STRIDE - 1) / DB_STRIDE, B);
    const int *run_if = nullptr;
    const char *n4 = "dblock_f4", *n8 = "dblock_f8";
    const Pipe pipe = fd_pipe(c, c->conv_f16 && w.dblock_f16_ok, 13 + d);
    if (pipe != PIPE_F32_ONLY) {
        int *flag = c->ws.range_flag + 13 + d;
        const float4 *q0 = reinterpret_cast<const float4 *>(w.down_h2[d][0]), *q1 = reinterpret_cast<const float4 *>(w.down_h2[d][1]),
                     *q2 = reinterpret_cast<const float4 *>(w.down_h2[d][2]), *q3 = reinterpret_cast<const float4 *>(w.down_h2[d][3]);
        const float *none = nullptr;
        if (f == 4 && d == 0 && audio)      // a[0] = first_audio_conv(audio): recomputed at the picked columns, not read
            FD_LAUNCH(L, n4, (k_dblock_h2<4, true>), grid, dim3(256), 0, c->ws.a[d], c->ws.a[d + 1], q0, q1, q2, q3, w.down[d].conv[0].b,
                      w.down[d].conv[1].b, w.down[d].conv[2].b, w.down[d].res.b, Lin, Lo, flag, c->step_lens, Lo / T, audio,
                      (const float *)w.first.w, (const float *)w.first.b);
        else if (f == 4)
            FD_LAUNCH(L, n4, (k_dblock_h2<4, false>), grid, dim3(256), 0, c->ws.a[d], c->ws.a[d + 1], q0, q1, q2, q3, w.down[d].conv[0].b,
                      w.down[d].conv[1].b, w.down[d].conv[2].b, w.down[d].res.b, Lin, Lo, flag, c->step_lens, Lo / T, none, none, none);
        else
            FD_LAUNCH(L, n8, (k_dblock_h2<8, false>), grid, dim3(256), 0, c->ws.a[d], c->ws.a[d + 1], q0, q1, q2, q3, w.down[d].conv[0].b,
                      w.down[d].conv[1].b, w.down[d].conv[2].b, w.down[d].res.b, Lin, Lo, flag, c->step_lens, Lo / T, none, none, none);
        run_if = flag;
        n4 = n8 = "dblock_fp32_fallback";
        if (pipe == PIPE_F16_ONLY) return hipSuccess;
    }
    if (f == 4)
        FD_LAUNCH(L, n4, k_dblock<4>, grid, dim3(256), 0, c->ws.a[d], c->ws.a[d + 1], w.down_pack[d][0], w.down_pack[d][1],
                  w.down_pack[d][2], w.down_pack[d][3], w.down[d].conv[0].b, w.down[d].conv[1].b, w.down[d].conv[2].b,
                  w.down[d].res.b, Lin, Lo, run_if, c->step_lens, Lo / T);
    else
        FD_LAUNCH(L, n8, k_dblock<8>, grid, dim3(256), 0, c->ws.a[d], c->ws.a[d + 1], w.down_pack[d][0], w.down_pack[d][1],
                  w.down_pack[d][2], w.down_pack[d][3], w.down[d].conv[0].b, w.down[d].conv[1].b, w.down[d].conv[2].b,
                  w.down[d].res.b, Lin, Lo, run_if, c->step_lens, Lo / T);
    return hipSuccess;
}

}  // namespace fdk
