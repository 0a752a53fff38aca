// fd_kernels_kp.hip -- a5 KernelPredictor: front (input conv + residual stack) and the kernel_conv / bias_conv GEMM (modules.py:257-343)
// (one stage of the gfx950 kernel set; shared device helpers: fd_kernels_common.h; the one-thread-per-output twins: fd_kernels_naive.hip)
#include "fd_kernels_common.h"

namespace fdk_fast {

// =================================================================================================
// a5 (front), fused: input conv + the six residual convs + skip add in ONE launch for all three predictors.
// Workgroup = one (block, utterance, 48-frame tile).  All seven layers are evaluated on the same 64 columns
// (frames t0-8 .. t0+55) with the activations ping-ponging through LDS; each k3 layer invalidates one column per side
// (the k5 input conv is covered by the staged +-2 halo), so columns 8..55 are exact after layer 7.  Wave = (32-row
// tile, 32-column tile).  Activations outside the utterance are forced to zero after every layer: that is the zero
// padding each reference conv applies to its own input.
// -------------------------------------------------------------------------------------------------
constexpr int KPF_VALID = 48, KPF_LDI = 68, KPF_LDH = 66;

struct KpFrontW {
    const float *in_pack[fd::NBLK], *in_b[fd::NBLK];
    const float *res_pack[fd::NBLK][6], *res_b[fd::NBLK][6];
};

__global__ void __launch_bounds__(256, 2) k_kp_front(const float *__restrict__ mel, float *__restrict__ hout, KpFrontW w,
                                                     const float *__restrict__ noise, const StepParams *params, int sampler,
                                                     int B, int T, const int *__restrict__ run_if, const int *__restrict__ lens)
{
    __shared__ float xin[fd::COND * KPF_LDI];     // mel + noise, columns <-> frames t0-10 .. t0+57
    if (run_if && *run_if == 0) return;           // fallback launch behind k_kp_front_h2
    __shared__ float h0[fd::HID * KPF_LDH];       // input-conv output (kept for the skip add), column c at index c+1
    __shared__ float hA[fd::HID * KPF_LDH];
    __shared__ float hB[fd::HID * KPF_LDH];
    const int blk = blockIdx.z, b = blockIdx.y, t0 = blockIdx.x * KPF_VALID;
    const int Tb = frames_of(lens, b, T);
    if (t0 >= Tb) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int mt = wave & 1, nt = wave >> 1;
    // sampler = 0: batch entry b reads row b (fd_forward; a hoisted whole call: row = step * B' + utterance already);  1: row
    // step_idx * B + b;  np >= 2 (the predictor of an np-step piece of a long schedule, B = np * B' entries): step_idx * B' + b
    const int step = sampler ? params->step_idx : 0;
    const float *nz = noise + (((int64_t)step * (sampler > 1 ? B / sampler : B) + b) * fd::NBLK + blk) * fd::COND;
    {   // stage mel + noise (loads batched), zero the guard columns of the activation buffers
        constexpr int TOTAL = fd::COND * KPF_LDI, NK = (TOTAL + 255) / 256;
        float v[NK];
        const float *src = mel + (int64_t)b * fd::COND * T;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid, ci = idx / KPF_LDI, cc = idx - ci * KPF_LDI, t = t0 - 10 + cc;
            v[k] = (idx < TOTAL && t >= 0 && t < Tb) ? src[(int64_t)ci * T + t] + nz[ci] : 0.0f;   // padding stays zero (modules.py:203)
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid;
            if (idx < TOTAL) xin[idx] = v[k];
        }
        if (tid < 128) {
            const int row = tid >> 1, col = (tid & 1) ? KPF_LDH - 1 : 0;
            h0[row * KPF_LDH + col] = 0.0f; hA[row * KPF_LDH + col] = 0.0f; hB[row * KPF_LDH + col] = 0.0f;
        }
    }
    __syncthreads();
    const int c = nt * 32 + l31;                 // this lane's column; frame t0 - 8 + c
    const int t = t0 - 8 + c;
    const bool inside = (t >= 0 && t < Tb);
    // ---- layer 0: Conv1d(80,64,k5,pad2) + lrelu 0.1 -------------------------------------------------------------------
    {
        const float4 *pa = reinterpret_cast<const float4 *>(w.in_pack[blk]) + (int64_t)mt * 50 * 64 + lane;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = w.in_b[blk][mt * 32 + drow(r, hi)];
#pragma unroll 5
        for (int s4 = 0; s4 < 50; ++s4) {
            const float4 a4 = pa[s4 * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 8 * s4 + 2 * r, tap = kk / fd::COND, ci = kk % fd::COND + hi;     // kk = tap*80 + ci
                acc = mfma32(f4c(a4, r), xin[ci * KPF_LDI + c + tap], acc);                        // frame t + tap - 2 -> column c + tap
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) h0[(mt * 32 + drow(r, hi)) * KPF_LDH + c + 1] = inside ? lrelu(acc[r], 0.1f) : 0.0f;
    }
    __syncthreads();
    // ---- six Conv1d(64,64,k3,pad1) + lrelu 0.1; the last one adds h0 and goes to HBM ----------------------------------
    const float *src = h0;
#pragma unroll 1
    for (int l = 0; l < 6; ++l) {
        float *dst = (l & 1) ? hB : hA;
        const float4 *pa = reinterpret_cast<const float4 *>(w.res_pack[blk][l]) + (int64_t)mt * 24 * 64 + lane;
        float4 wa[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) wa[i] = pa[i * 64];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = w.res_b[blk][l][mt * 32 + drow(r, hi)];
#pragma unroll
        for (int s = 0; s < 96; ++s) {           // kk = 2s+hi = tap*64 + ci
            const int tap = s >> 5, ci = ((2 * s) & 63) + hi;
            acc = mfma32(f4c(wa[s >> 2], s & 3), src[ci * KPF_LDH + c + tap], acc);               // column c + tap - 1 at index c + tap
        }
        if (l < 5) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(mt * 32 + drow(r, hi)) * KPF_LDH + c + 1] = inside ? lrelu(acc[r], 0.1f) : 0.0f;
            __syncthreads();
            src = dst;
        } else if (inside && c >= 8 && c < 8 + KPF_VALID) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = mt * 32 + drow(r, hi);
                hout[(((int64_t)blk * B + b) * fd::HID + o) * T + t] = lrelu(acc[r], 0.1f) + h0[o * KPF_LDH + c + 1];
            }
        }
    }
}

// The same seven layers on the fp16 matrix pipe with 2-piece operands: 291 MFMAs of 32 cycles per
// wave instead of 776 of 64.  Images: mel + noise as [column][piece][80 ch] fp16, 336 B per column (320 + 16: a row stride of
// 84 dwords = 4 x odd spreads the 16-lane groups of ds_read_b128 over all banks without a swizzle); activations as
// [column + 1][piece][64 ch], 256 B per column, slots swizzled by row & 15 -- the layout of the GEMM's h image, which the last
// layer writes directly (k_h_split is not needed behind this kernel) next to the fp32 h the fallback kernels and the taps read.
constexpr int KPF_XROW = 336;

struct KpFrontW2 {
    const float4 *in_pack[fd::NBLK];
    const float *in_b[fd::NBLK];
    const float4 *res_pack[fd::NBLK][6];
    const float *res_b[fd::NBLK][6];
};

__device__ __forceinline__ int kpf_off(int row, int slot) { return row * 256 + ((slot ^ (row & 15)) << 4); }

__global__ void __launch_bounds__(256, 2) k_kp_front_h2(const float *__restrict__ mel, float *__restrict__ hout, char *__restrict__ himg,
                                                        KpFrontW2 w, const float *__restrict__ noise, const StepParams *params, int sampler,
                                                        int B, int T, int R, int *__restrict__ range_flags, const int *__restrict__ lens)
{
    __shared__ __attribute__((aligned(16))) char xin[68 * KPF_XROW];      // columns <-> frames t0-10 .. t0+57
    __shared__ __attribute__((aligned(16))) char hA[66 * 256];            // column c at row c+1; rows 0 and 65 stay zero
    __shared__ __attribute__((aligned(16))) char hB[66 * 256];
    const int blk = blockIdx.z, b = blockIdx.y, t0 = blockIdx.x * KPF_VALID;
    const int Tb = frames_of(lens, b, T);
    if (range_flags[32 + 19] | range_flags[32]) {      // did not fit in the previous step: fp32 front and fp32 GEMM take this one
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { atomicOr(range_flags + 19, 1); atomicOr(range_flags, 1); }
        return;
    }
    if (t0 > Tb) return;      // the tile holding frame Tb still runs: it writes the zero row the GEMM reads behind the utterance
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int mt = wave & 1, nt = wave >> 1;
    // sampler = 0: batch entry b reads row b (fd_forward; a hoisted whole call: row = step * B' + utterance already);  1: row
    // step_idx * B + b;  np >= 2 (the predictor of an np-step piece of a long schedule, B = np * B' entries): step_idx * B' + b
    const int step = sampler ? params->step_idx : 0;
    const float *nz = noise + (((int64_t)step * (sampler > 1 ? B / sampler : B) + b) * fd::NBLK + blk) * fd::COND;
    float mx = 0.0f;
    {   // stage mel + noise: thread = (8-channel group of 10, column of 68) = 680 units; padding stays zero (modules.py:203)
        const float *src = mel + (int64_t)b * fd::COND * T;
        float v[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = k * 256 + tid, cg = u / 68, cc = u - cg * 68, t = t0 - 10 + cc;
            const bool ok = u < 680 && t >= 0 && t < Tb;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[k][c] = ok ? src[(int64_t)(cg * 8 + c) * T + t] + nz[cg * 8 + c] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = k * 256 + tid, cg = u / 68, cc = u - cg * 68;
            if (u < 680) {
#pragma unroll
                for (int c = 0; c < 8; ++c) mx = fmaxf(mx, fabsf(v[k][c]));
                float4 ph, pl;
                split8(v[k], ph, pl);
                *reinterpret_cast<float4 *>(xin + cc * KPF_XROW + cg * 16) = ph;
                *reinterpret_cast<float4 *>(xin + cc * KPF_XROW + 160 + cg * 16) = pl;
            }
        }
        if (tid < 64) {         // guard rows 0 and 65 of both activation images
            const int row = (tid & 32) ? 65 : 0, part = tid & 15;
            char *img = (tid & 16) ? hB : hA;
            *reinterpret_cast<float4 *>(img + row * 256 + part * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    const int c = nt * 32 + l31;                 // this lane's column; frame t0 - 8 + c
    const int t = t0 - 8 + c;
    const bool inside = (t >= 0 && t < Tb);
    float h0v[16];                               // this lane's layer-0 outputs (fp32) for the skip add of the last layer
    // write leaky_relu(hi + 2^-11 lo) (0 outside the utterance) as pieces of column c: D rows 32*mt + 8j + 4hi + {0..3}
    auto store_act = [&](char *img, const f32x16 &ah, const f32x16 &al, float *keep) {
        const int row = c + 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = inside ? lrelu(fmaf(al[4 * j + i], GX_INV_SCALE, ah[4 * j + i]), 0.1f) : 0.0f;
                mx = fmaxf(mx, fabsf(v[i]));
                if (keep) keep[4 * j + i] = v[i];
            }
            uint2 ph, pl;
            split2(v[0], v[1], ph.x, pl.x);
            split2(v[2], v[3], ph.y, pl.y);
            *reinterpret_cast<uint2 *>(img + kpf_off(row, mt * 4 + j) + 8 * hi) = ph;
            *reinterpret_cast<uint2 *>(img + kpf_off(row, 8 + mt * 4 + j) + 8 * hi) = pl;
        }
    };
    // ---- layer 0: Conv1d(80,64,k5,pad2) + lrelu 0.1; k = 16*kg + 8*hi + e = tap*80 + ci, weights streamed from L2 --------------
    {
        const float4 *pa = w.in_pack[blk] + (int64_t)mt * 2 * 25 * 64 + lane;
        f32x16 ah, al;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ah[r] = w.in_b[blk][mt * 32 + drow(r, hi)]; al[r] = 0.0f; }
        const char *xb = xin + c * KPF_XROW + hi * 16;
#pragma unroll 5
        for (int kg = 0; kg < 25; ++kg) {
            const float4 w1 = pa[kg * 64], w2 = pa[(25 + kg) * 64];
            const int tap = (16 * kg) / fd::COND, o = ((16 * kg) % fd::COND) / 8;      // frame t + tap - 2 -> column c + tap
            const float4 b1 = *reinterpret_cast<const float4 *>(xb + tap * KPF_XROW + o * 16);
            const float4 b2 = *reinterpret_cast<const float4 *>(xb + tap * KPF_XROW + 160 + o * 16);
            ah = mfma_f16(w1, b1, ah);
            al = mfma_f16(w1, b2, al);
            al = mfma_f16(w2, b1, al);
        }
        store_act(hA, ah, al, h0v);
    }
    __syncthreads();
    // ---- six Conv1d(64,64,k3,pad1) + lrelu 0.1; the last one adds the layer-0 output and goes to HBM ------------------------
    int off[3][2][4];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) off[tap][p][k4] = kpf_off(c + tap, p * 8 + k4 * 2 + hi);      // column c + tap - 1 at row c + tap
    // the weights of layer l+1 are requested as soon as the MFMAs of layer l are issued: their L2 latency then hides behind
    // the activation split / LDS write-back / barrier of layer l instead of stalling the next layer
    float4 wa[2][12];
    auto load_w = [&](int l) {
        const float4 *pa = w.res_pack[blk][l] + (int64_t)mt * 2 * 12 * 64 + lane;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int kg = 0; kg < 12; ++kg) wa[p][kg] = pa[(p * 12 + kg) * 64];
    };
    load_w(0);
#pragma unroll 1
    for (int l = 0; l < 6; ++l) {
        const char *src = (l & 1) ? hB : hA;
        char *dst = (l & 1) ? hA : hB;
        f32x16 ah, al;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ah[r] = w.res_b[blk][l][mt * 32 + drow(r, hi)]; al[r] = 0.0f; }
#pragma unroll
        for (int kg = 0; kg < 12; ++kg) {           // k = 16*kg + 8*hi + e = tap*64 + ci
            const float4 b1 = *reinterpret_cast<const float4 *>(src + off[kg >> 2][0][kg & 3]);
            const float4 b2 = *reinterpret_cast<const float4 *>(src + off[kg >> 2][1][kg & 3]);
            ah = mfma_f16(wa[0][kg], b1, ah);
            al = mfma_f16(wa[0][kg], b2, al);
            al = mfma_f16(wa[1][kg], b1, al);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (l < 5) load_w(l + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (l < 5) {
            store_act(dst, ah, al, nullptr);
            __syncthreads();
        } else if (t >= 0 && t < T && c >= 8 && c < 8 + KPF_VALID) {      // frames in [Tb, T) are written as zeros
            char *irow = himg + (((int64_t)blk * B + b) * R + (t + 1)) * 256;       // the GEMM's image: row = frame + 1
            const int sw = (t + 1) & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * j + i;
                    v[i] = inside ? lrelu(fmaf(al[r], GX_INV_SCALE, ah[r]), 0.1f) + h0v[r] : 0.0f;
                    mx = fmaxf(mx, fabsf(v[i]));
                    hout[(((int64_t)blk * B + b) * fd::HID + mt * 32 + drow(r, hi)) * T + t] = v[i];
                }
                if (himg) {      // (null: the GEMM builds its own operand image from hout -- the Winograd form, k_h_wino)
                    uint2 ph, pl;
                    split2(v[0], v[1], ph.x, pl.x);
                    split2(v[2], v[3], ph.y, pl.y);
                    *reinterpret_cast<uint2 *>(irow + (((mt * 4 + j) ^ sw) << 4) + 8 * hi) = ph;
                    *reinterpret_cast<uint2 *>(irow + (((8 + mt * 4 + j) ^ sw) << 4) + 8 * hi) = pl;
                }
            }
        }
    }
    // the image's padding rows (0 and T+1 .. R-1) must read as zeros: first and last tile of the utterance write them
    if (himg) {
        char *ib = himg + ((int64_t)blk * B + b) * R * 256;
        if (blockIdx.x == 0 && tid < 16) *reinterpret_cast<float4 *>(ib + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (blockIdx.x == gridDim.x - 1)
            for (int e = tid; e < (R - 1 - T) * 16; e += 256) *reinterpret_cast<float4 *>(ib + (T + 1) * 256 + e * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!(mx < GX_LIMIT)) { atomicOr(range_flags + 19, 1); atomicOr(range_flags, 1); }      // [19] this kernel, [0] the GEMM behind it
}

// =================================================================================================
// a5 (GEMM): kernel_conv + bias_conv (modules.py:315-318,330-331) as ONE fp32-MFMA GEMM per LVC block:
//   kpack[b][t][p] = gbias[p] + sum_{tap,c} Wp[p][tap*64+c] * h[b][c][t+tap-1],   p in [0,24832)
// rows of the MFMA = frames (A from an LDS window of h), cols = 32 consecutive packed positions p
// (B = weights, register-stationary: 96 VGPRs per wave, loaded once and reused for every frame tile of the
// workgroup's chunk).  Output goes out frame-major so the LVC kernel reads a frame's record contiguously.
// =================================================================================================
constexpr int GEMM_CT = 4;                       // frame tiles (of 32) per work item
constexpr int GEMM_LDH = GEMM_CT * 32 + 4;       // 128 frames + 1 halo each side, padded
constexpr int GEMM_NCOLS = GEMM_CT * 32 + 2;
constexpr int GEMM_NK = (fd::HID * GEMM_NCOLS + 255) / 256;   // staged floats per thread

#ifdef FD_GEMM_TIMING
__device__ long long fd_gdbg[64 * 4 * 4];
#endif

// Persistent, software-pipelined form.  A work item = (LVC block, 128-column group, utterance, chunk of <= 4 frame tiles).
// Each of the 2 x #CU workgroups owns a CONTIGUOUS range of items, ordered so that consecutive items share the column
// group: the 96 weight registers of a wave are re-loaded only when the group changes (about once per workgroup), the h
// window of item i+1 is fetched into registers before the MFMAs of item i and written to the other LDS buffer after
// them, and the only barrier is one per item.  The matrix pipe never waits on a prologue.
__global__ void __launch_bounds__(256, 2) k_kp_gemm(const float *__restrict__ h /*[3][B][64][T]*/, float *__restrict__ kpack,
                                                    const float *g0, const float *g1, const float *g2, const float *gb0,
                                                    const float *gb1, const float *gb2, int B, int T, int chunks_per_utt,
                                                    int chunk_tiles, int n_items, const int *__restrict__ run_if, const int *__restrict__ lens)
{
    __shared__ float hs[2][fd::HID * GEMM_LDH];
    if (run_if && *run_if == 0) return;      // fallback launch behind the fp16 kernel: only when k_h_split flagged the operands
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    constexpr int XG = fd::KREC / 128;
    const int ny = B * chunks_per_utt;
    const int i0 = (int)((int64_t)blockIdx.x * n_items / gridDim.x), i1 = (int)((int64_t)(blockIdx.x + 1) * n_items / gridDim.x);
    if (i0 >= i1) return;

    struct Item { int blk, xg, b, t_begin; };
    auto decode = [&](int id) {
        Item it;
        it.blk = id / (XG * ny);
        const int rem = id - it.blk * (XG * ny);
        it.xg = rem / ny;
        const int yy = rem - it.xg * ny;
        it.b = yy / chunks_per_utt;
        it.t_begin = (yy - it.b * chunks_per_utt) * chunk_tiles * 32;
        return it;
    };
    // h[blk][b][:, t_begin-1 .. t_begin+128] -> registers (zero outside the utterance) -> LDS buffer.  Thread = (column
    // tid%128, row parity tid/128): every per-load address is base + j*const, so nothing per-element stays live (a flat
    // idx/130 mapping makes the compiler keep ~100 hoisted offsets in registers and spill them).
    float v[33];
    const int scol = tid & 127, srow = tid >> 7;
#define FD_GEMM_FETCH(it)                                                                                              \
    do {                                                                                                               \
        const float *hb__ = h + (((int64_t)(it).blk * B + (it).b) * fd::HID + srow) * T + ((it).t_begin - 1);          \
        const bool ok__ = ((it).t_begin - 1 + scol) >= 0 && ((it).t_begin - 1 + scol) < frames_of(lens, (it).b, T);    \
        _Pragma("unroll") for (int j = 0; j < 32; ++j) v[j] = ok__ ? hb__[(int64_t)(2 * j) * T + scol] : 0.0f;         \
        const int t2__ = (it).t_begin + 127 + (tid & 1);     /* columns 128,129 of rows 0..63: threads 0..127 */          \
        v[32] = (tid < 128 && t2__ < frames_of(lens, (it).b, T)) ? hb__[(int64_t)((tid >> 1) - srow) * T + 128 + (tid & 1)] : 0.0f;                \
    } while (0)
#define FD_GEMM_COMMIT(bufi)                                                                                           \
    do {                                                                                                               \
        float *hd__ = hs[bufi] + srow * GEMM_LDH + scol;                                                               \
        _Pragma("unroll") for (int j = 0; j < 32; ++j) hd__[2 * j * GEMM_LDH] = v[j];                                   \
        if (tid < 128) hs[bufi][(tid >> 1) * GEMM_LDH + 128 + (tid & 1)] = v[32];                                       \
    } while (0)

    Item cur = decode(i0);
    FD_GEMM_FETCH(cur);
    FD_GEMM_COMMIT(0);
    __syncthreads();
    float4 wb[24];
    float bias = 0.0f;
    int have_blk = -1, have_xg = -1, buf = 0;
#pragma unroll 1
    for (int i = i0; i < i1; ++i) {
        if (cur.blk != have_blk || cur.xg != have_xg) {      // new column group: (re)load the register-stationary weights
            const float *gp = cur.blk == 0 ? g0 : (cur.blk == 1 ? g1 : g2);
            const float *gb = cur.blk == 0 ? gb0 : (cur.blk == 1 ? gb1 : gb2);
            const int ptile = cur.xg * 4 + wave;
#pragma unroll
            for (int k = 0; k < 24; ++k) wb[k] = reinterpret_cast<const float4 *>(gp)[((int64_t)ptile * 24 + k) * 64 + lane];
            bias = gb[ptile * 32 + l31];
            have_blk = cur.blk; have_xg = cur.xg;
        }
        Item nxt = cur;
        const bool more = (i + 1 < i1);
        if (more) { nxt = decode(i + 1); FD_GEMM_FETCH(nxt); }
        const int Tb = frames_of(lens, cur.b, T);
        const int n_frames = min(Tb - cur.t_begin, chunk_tiles * 32);
        const int n_tiles = (n_frames + 31) >> 5;
        float *kout = kpack + ((int64_t)cur.blk * B + cur.b) * T * fd::KREC + (cur.xg * 4 + wave) * 32 + l31;   // + t*KREC
#pragma unroll 1
        for (int tile = 0; tile < n_tiles; ++tile) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias;
            const int hc = hi * GEMM_LDH + l31 + tile * 32;
            const int ho3[3] = {opaque(hc), opaque(hc + 1), opaque(hc + 2)};
#pragma unroll
            for (int s = 0; s < 96; ++s) {     // kk = 2s+hi = tap*64 + c ; frame column = local frame + tap (column 0 is t_begin-1)
                const int tap = s >> 5, c2 = (2 * s) & 63;
                acc = mfma32(hs[buf][ho3[tap] + c2 * GEMM_LDH], f4c(wb[s >> 2], s & 3), acc);
            }
            const int t0 = cur.t_begin + tile * 32;
            const unsigned base = (unsigned)(t0 + 4 * hi) * (unsigned)fd::KREC;     // < 2^32: checked on the host
#ifdef FD_GX_NO_STORE
            if (acc[0] != 12345.678f) continue;
#endif
            if (t0 + 32 <= Tb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) kout[base + (unsigned)(((r & 3) + 8 * (r >> 2)) * fd::KREC)] = acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t0 + drow(r, hi) < Tb) kout[base + (unsigned)(((r & 3) + 8 * (r >> 2)) * fd::KREC)] = acc[r];
            }
        }
        if (more) FD_GEMM_COMMIT(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        cur = nxt;
    }
}

// -------------------------------------------------------------------------------------------------
// ---- split-precision form on the fp16 matrix pipe ------------------------------------------------------------------------
// x = x1 + 2^-11 * x2 with x1 = fp16(x), x2 = fp16((x - x1) * 2^11): 22 significant bits per operand.  W.h is evaluated as
//   hi = W1.h1 (fp32 accumulate)     lo = W1.h2 + W2.h1 (separate fp32 accumulator)     result = bias + hi + 2^-11 * lo
// three v_mfma_f32_32x32x16_f16 per 16 k (32 cycles each) instead of eight v_mfma_f32_32x32x2f32 (64 cycles each).  The
// neglected W2.h2 term is 2^-22 relative, the same order as the representation error; measured against a float64 product
// the result is closer than an fp32 sgemm.  fp16 subnormals are honoured by v_cvt and by the MFMA
// (tools/ubench/f16_probe.hip), so small values lose nothing; operands of magnitude >= 32768 do not fit: k_h_split raises
// a flag for them, this kernel then leaves the step to the fp32 kernel that follows it in the stream.
constexpr int GX_CT = 4;                        // frame tiles per item
constexpr int GX_ROWS = GX_CT * 32 + 2;         // 130 rows: frames t_begin-1 .. t_begin+128
constexpr int GX_ROWB = 2 * 128;                // bytes per row of the piece image: [piece][64 ch] fp16
constexpr int GX_WINB = GX_ROWS * GX_ROWB;      // 33280 B per item window
constexpr int GX_NDMA = (GX_WINB + 4095) / 4096;   // 4 KB (256 lanes x 16 B) DMA rounds per window: 8 full + 1 partial
constexpr int GX_BUFB = GX_NDMA * 4096;         // LDS bytes per buffer (the partial round is padded to a whole wave)
// Cache policy bits of the predicted-kernel stores (1 = sc0, 2 = nt, 16 = sc1).  nt since round 5: the 2 GB of records stream past L2
// instead of turning it over under the window reads, and fewer dirty lines are left for the hop-8 layers behind this kernel to
// wait on -- A/B in one session on a box of the slower kind (profiles/r05/s1_gemm_store_policy.txt): kernel 706 -> 567 us, the first
// hop-8 layer 82.8 -> 65.0 us, a reverse step 2575 -> 2416 us; sc0 sc1 (write-through) 671 us; nt on block 0's records only 582 us.
#ifndef FD_GX_STORE_AUX
#define FD_GX_STORE_AUX 2
#endif

__host__ __device__ inline int gx_rows(int T) { return ((T + GX_CT * 32 - 1) / (GX_CT * 32)) * (GX_CT * 32) + 2; }   // image rows per (block, utterance)


// h (fp32 [3][B][64][T]) -> fp16 piece image [3][B][row = t+1][piece][64 channels]; rows 0 and > T are zero.
// A row is 256 B = 16 slots of 16 B; slot s of row r is stored at s ^ (r & 15): the 16-lane service groups of ds_read_b128
// (rows l, l+1, ... of one slot) then touch every bank once, and because the swizzle depends only on the ABSOLUTE row
// (item windows start at multiples of 64 frames) the GEMM can pull a window into LDS as one linear DMA copy.
__global__ void __launch_bounds__(256) k_h_split(const float *__restrict__ h, unsigned *__restrict__ hx, int *__restrict__ range_flag,
                                                 int B, int T, int R, const int *__restrict__ lens)
{
    const int bb = blockIdx.y;                           // blk*B + b
    const int e = blockIdx.x * 256 + threadIdx.x, cp = e / R, row = e - cp * R;     // lanes along rows: coalesced h reads
    if (cp >= 32) return;
    if (range_flag[32] != 0) {      // h did not fit in the previous step: the fp32 GEMM takes this one as well
        if (e == 0 && bb == 0) atomicOr(range_flag, 1);
        return;
    }
    const int t = row - 1;
    const bool ok = t >= 0 && t < frames_of(lens, bb % B, T);
    const float *hb = h + (int64_t)bb * fd::HID * T;
    const float a = ok ? hb[(int64_t)(2 * cp) * T + t] : 0.0f, b2 = ok ? hb[(int64_t)(2 * cp + 1) * T + t] : 0.0f;
    if (!(fmaxf(fabsf(a), fabsf(b2)) < GX_LIMIT)) atomicOr(range_flag, 1);      // also catches NaN / inf
    const _Float16 a1 = (_Float16)a, b1 = (_Float16)b2;
    const _Float16 a2 = (_Float16)((a - (float)a1) * GX_SCALE), b3 = (_Float16)((b2 - (float)b1) * GX_SCALE);
    union { _Float16 h[2]; unsigned u; } p1, p2;
    p1.h[0] = a1; p1.h[1] = b1;
    p2.h[0] = a2; p2.h[1] = b3;
    unsigned *dst = hx + ((int64_t)bb * R + row) * 64 + (cp & 3);
    const unsigned sw = (unsigned)row & 15u, slot = (unsigned)cp >> 2;
    dst[((slot ^ sw) << 2)] = p1.u;
    dst[(((slot + 8u) ^ sw) << 2)] = p2.u;
}

typedef __attribute__((address_space(3))) void *lds_ptr_t;

struct GxItem { int blk, xg, b, chunk; };

// Async copy of one item window (16896 B, contiguous in the piece image) into an LDS buffer: 16 B per lane, LDS side linear
// (M0 = wave-uniform LDS base, lane i lands at base + 16*i).  Issued as inline asm on purpose: for the builtin the compiler
// puts a full vmcnt(0) in front of the next ds_read of ANY LDS address, which would serialise the copy with the MFMAs of the
// current item; the waits are counted by hand in gx_item instead.
__device__ __forceinline__ void gx_dma(const char *hx, char *lds_buf, const GxItem &it, int B, int R, int wave_u, int lane)
{
#ifndef FD_GX_NO_FETCH
    const char *src = hx + (((int64_t)it.blk * B + it.b) * R + it.chunk * (GX_CT * 32)) * GX_ROWB + wave_u * 1024;    // uniform
    const unsigned dst = (unsigned)(uintptr_t)(lds_ptr_t)(lds_buf + wave_u * 1024);
    const unsigned voff = lane * 16;
    unsigned keep;
#pragma unroll
    for (int j = 0; j < GX_NDMA; ++j) {
        if (j == GX_NDMA - 1 && wave_u != 0) break;      // the last 512 B (rounded to one wave; the image has slack behind it)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(src + j * 4096), "s"(dst + j * 4096)
                     : "memory");
    }
#endif
}

// One work item = (LVC block, 128-column group, utterance, 64 frames): 2 frame tiles x 36 MFMAs per wave.
//   BUF   which LDS buffer holds this item's window (the other one receives the next item's window by DMA meanwhile)
//   FULL  both tiles are whole (always, except the ragged last chunk of an utterance)
// Vector-memory order per item: [DMA of next window] [16 stores of tile 0] [16 stores of tile 1].  vmcnt retires in order,
// so "vmcnt(32)" at the end of the item waits for the DMA (and the previous item's stores, a whole item old by then) but
// not for this item's stores.
#ifdef FD_GX_TIMING
__device__ long long fd_gxdbg[8];
#define GX_STAMP(k) do { const long long t__ = __builtin_amdgcn_s_memtime(); ph[k] += t__ - tl; tl = t__; } while (0)
#define GX_TIMING_ARGS , long long (&ph)[8], long long &tl
#define GX_TIMING_PASS , ph, tl
#else
#define GX_STAMP(k) do { } while (0)
#define GX_TIMING_ARGS
#define GX_TIMING_PASS
#endif
template <int BUF, bool FULL>
__device__ __forceinline__ void gx_item(char *lds, const GxItem &cur, bool more, const GxItem &nxt, const char *hx, float *kpack,
                                        const float4 (&wq)[2][12], const f32x16 &bias_lo, const int (&aoff)[2][12], int B, int T, int R,
                                        int wave_u, int lane, int Tb GX_TIMING_ARGS)
{
    const int l31 = lane & 31, hi = lane >> 5;
    GX_STAMP(0);
    if (more) gx_dma(hx, lds + (BUF ^ 1) * GX_BUFB, nxt, B, R, wave_u, lane);
    GX_STAMP(1);
    const int t_begin = cur.chunk * (GX_CT * 32);
    float *krow = kpack + (((int64_t)cur.blk * B + cur.b) * T + t_begin) * fd::KREC + (cur.xg * 4 + wave_u) * 32;     // uniform
    const unsigned loff = (unsigned)(4 * hi) * (unsigned)fd::KREC + (unsigned)l31;
    // stores of whole tiles go through a buffer descriptor: address = base (SGPRs) + per-lane offset (one VGPR, constant) + row
    // offset (an SGPR literal), so that a store costs no VALU instruction next to the MFMAs of the other wave on this SIMD
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(krow, 0, GX_CT * 32 * fd::KREC * 4, 0x00020000);
    const int n_tiles = max(0, min(GX_CT, (Tb - t_begin + 31) >> 5));      // Tb: frames of this utterance; FULL: all whole tiles
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tile = 0; tile < GX_CT; ++tile) {
        if (tile >= n_tiles) break;
        const char *hb = lds + BUF * GX_BUFB + tile * 32 * GX_ROWB;
        // kg = tap*4 + k4: logical k = kg*16 + 8*hi + e = tap*64 + channel.  The operands of step kg+1 are requested before
        // the MFMAs of step kg.  acc starts from 0 (inline constant), lo from 2048*bias: the bias then comes out of the final fma.
        f32x16 acc, lo;
        float4 a1 = *reinterpret_cast<const float4 *>(hb + aoff[0][0]), a2 = *reinterpret_cast<const float4 *>(hb + aoff[1][0]);
#pragma unroll
        for (int kg = 0; kg < 12; ++kg) {
            float4 n1 = a1, n2 = a2;
            if (kg + 1 < 12) {
                n1 = *reinterpret_cast<const float4 *>(hb + aoff[0][kg + 1]);
                n2 = *reinterpret_cast<const float4 *>(hb + aoff[1][kg + 1]);
            }
            acc = mfma_f16(a1, wq[0][kg], kg == 0 ? zero : acc);
            lo = mfma_f16(a2, wq[0][kg], kg == 0 ? bias_lo : lo);
#ifndef FD_GX_PROBE_TWO_MFMA   // probe (round 6): two of the three matrix instructions per k group -- what would a third less matrix work buy?
            lo = mfma_f16(a1, wq[1][kg], lo);
#endif
            a1 = n1;
            a2 = n2;
        }
        GX_STAMP(2);
#ifdef FD_GX_NO_STORE
        if (acc[0] != 12345.678f) continue;
#endif
        if (FULL) {
#ifdef FD_GX_STORE_AUX_B0     // probe (tools/gpu_r5_s1.sh): block 0's records -- the ones the hop-8 layers right behind this kernel read -- under a policy of their own
            if (cur.blk == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaf(lo[r], GX_INV_SCALE, acc[r])), rs, loff * 4u,
                                                          (tile * 32 + (r & 3) + 8 * (r >> 2)) * fd::KREC * 4, FD_GX_STORE_AUX_B0);
            } else
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaf(lo[r], GX_INV_SCALE, acc[r])), rs, loff * 4u,
                                                      (tile * 32 + (r & 3) + 8 * (r >> 2)) * fd::KREC * 4, FD_GX_STORE_AUX);
        } else {
            float *kt = krow + (int64_t)tile * 32 * fd::KREC;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t_begin + tile * 32 + drow(r, hi) < Tb)
                    (kt + ((r & 3) + 8 * (r >> 2)) * fd::KREC)[loff] = fmaf(lo[r], GX_INV_SCALE, acc[r]);
        }
        GX_STAMP(3);
    }
    if (more) {      // the DMA has landed; this item's 16 * n_tiles buffer stores may still fly
        if (FULL && n_tiles == 4) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");       // 6-bit counter: 63 is its ceiling
        else if (FULL && n_tiles == 3) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        else if (FULL && n_tiles == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (FULL && n_tiles == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    GX_STAMP(4);
    __builtin_amdgcn_s_barrier();      // every wave's DMA share is in LDS; everybody is done reading this item's buffer
    GX_STAMP(5);
}

__global__ void __launch_bounds__(256, 2) k_kp_gemm_h2(const char *__restrict__ hx /*[3][B][R][2][64] fp16*/, float *__restrict__ kpack,
                                                       const float4 *g0, const float4 *g1, const float4 *g2, const float *gb0,
                                                       const float *gb1, const float *gb2, const int *__restrict__ range_flag, int B,
                                                       int T, int R, int chunks_per_utt, int n_items, const int *__restrict__ lens,
                                                       int blk0, int nblk)
{
    // blk0, nblk: the LVC blocks this launch computes (0, 3: all of them; option overlap = gemm launches block 0 alone and the other
    // two next to the LVC layers of the block before them); n_items counts the items of those blocks only
    __shared__ __attribute__((aligned(16))) char lds[2 * GX_BUFB];     // 2 x 36 KB
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int XG = fd::KREC / 128;
    if (*range_flag != 0) return;      // out-of-range operands (raised by the producer of h): the fp32 kernel behind us does this step

    // Work items: id = ((block*XG + column group)*B + utterance)*chunks + chunk; a column group (128 columns, the weights a
    // workgroup keeps in registers) spans ny = B*chunks consecutive ids, one per 64-frame window of the block's h image.
    // Schedule: workgroup w first does whole groups w, w + #wg, ...: all workgroups then walk the windows in step, and a window
    // is fetched from HBM once per XCD instead of once per workgroup (the 2 GB output stream turns L2 over every few
    // microseconds).  The groups that do not divide evenly are cut into equal contiguous id ranges at the end.
    const int ny = B * chunks_per_utt, n_wg = gridDim.x, w = blockIdx.x;
    const int q = (nblk * XG) / n_wg, base = q * n_wg * ny, rest = n_items - base;
    const int r0 = (int)((int64_t)w * rest / n_wg), r1 = (int)((int64_t)(w + 1) * rest / n_wg);
    const int n_mine = q * ny + (r1 - r0);
    if (n_mine <= 0) return;
    auto decode = [&](int id) {
        GxItem it;
        it.blk = id / (XG * ny);
        const int rem = id - it.blk * (XG * ny);
        it.blk += blk0;
        it.xg = rem / ny;
        const int yy = rem - it.xg * ny;
        it.b = yy / chunks_per_utt;
        it.chunk = yy - it.b * chunks_per_utt;
        return it;
    };
    auto advance = [&](GxItem it) {
        if (++it.chunk == chunks_per_utt) {
            it.chunk = 0;
            if (++it.b == B) {
                it.b = 0;
                if (++it.xg == XG) { it.xg = 0; ++it.blk; }
            }
        }
        return it;
    };
    int run = 0, left = (q > 0) ? ny : (r1 - r0);          // ids of a run are consecutive; `left` counts the current item too
    GxItem cur = decode((q > 0) ? w * ny : base + r0);

    // byte offsets of the A-operand reads of a tile: row = frame + tap, slot = (8*piece + 2*k4 + hi) ^ (row & 15)
    int aoff[2][12];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int row = l31 + tap;
                aoff[q][tap * 4 + k4] = row * GX_ROWB + (((q * 8 + k4 * 2 + hi) ^ (row & 15)) << 4);
            }

    gx_dma(hx, lds, cur, B, R, wave_u, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#ifdef FD_GX_TIMING
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
    const long long t_start = tl, r_start = __builtin_amdgcn_s_memrealtime();
#endif
    float4 wq[2][12];
    f32x16 bias_lo;      // 2048 * bias of this lane's column in all 16 registers: the C operand of the first cross-term MFMA
    int have_blk = -1, have_xg = -1;
#pragma unroll 1
    for (int i = 0; i < n_mine; i += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && i + 1 >= n_mine) break;
            if (cur.blk != have_blk || cur.xg != have_xg) {      // new column group: (re)load the register-stationary weight pieces
                const float4 *gp = cur.blk == 0 ? g0 : (cur.blk == 1 ? g1 : g2);
                const float *gb = cur.blk == 0 ? gb0 : (cur.blk == 1 ? gb1 : gb2);
                const int ptile = cur.xg * 4 + wave_u;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int kg = 0; kg < 12; ++kg) wq[q2][kg] = gp[(((int64_t)ptile * 2 + q2) * 12 + kg) * 64 + lane];
                const float bv = gb[ptile * 32 + l31] * GX_SCALE;
#pragma unroll
                for (int r = 0; r < 16; ++r) bias_lo[r] = bv;
                have_blk = cur.blk; have_xg = cur.xg;
                // retire the loads here, visibly to the compiler: otherwise it places a vmcnt(0) at the first use, on the
                // common path too, where it would wait for the window DMA and the stores of the previous item
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            const bool more = (i + half + 1 < n_mine);
            GxItem nxt = cur;
            if (more) {
                if (left > 1) nxt = advance(cur);
                else {                                           // next run: the next whole group, or the tail range
                    ++run;
                    nxt = decode(run < q ? (run * n_wg + w) * ny : base + r0);
                    left = (run < q ? ny : r1 - r0) + 1;
                }
            }
            --left;
            const int Tb = frames_of(lens, cur.b, T);
            const bool full = (Tb % 32 == 0) || (cur.chunk * (GX_CT * 32) + GX_CT * 32 <= Tb);
            if (half == 0) {
                if (full) gx_item<0, true>(lds, cur, more, nxt, hx, kpack, wq, bias_lo, aoff, B, T, R, wave_u, lane, Tb GX_TIMING_PASS);
                else gx_item<0, false>(lds, cur, more, nxt, hx, kpack, wq, bias_lo, aoff, B, T, R, wave_u, lane, Tb GX_TIMING_PASS);
            } else {
                if (full) gx_item<1, true>(lds, cur, more, nxt, hx, kpack, wq, bias_lo, aoff, B, T, R, wave_u, lane, Tb GX_TIMING_PASS);
                else gx_item<1, false>(lds, cur, more, nxt, hx, kpack, wq, bias_lo, aoff, B, T, R, wave_u, lane, Tb GX_TIMING_PASS);
            }
            cur = nxt;
        }
    }
#ifdef FD_GX_TIMING
    ph[6] = __builtin_amdgcn_s_memtime() - t_start;
    ph[7] = __builtin_amdgcn_s_memrealtime() - r_start;
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long *)&fd_gxdbg[k], (unsigned long long)ph[k]);
#endif
}

// =================================================================================================
// The same product as Winograd F(2,3) over the FRAME axis (round 6).  kernel_conv is a k = 3 convolution over frames
// (modules.py:315-318), and k_kp_gemm_h2 is bound by its matrix instructions at the power-capped clock (LABBOOK R6.8: with two of
// its three instructions per k group it runs in 407 instead of 514 us).  Per pair of output frames (2p, 2p+1) four K = 64 products
// replace six:
//   u0 = h[2p-1] - h[2p+1]   u1 = h[2p] + h[2p+1]   u2 = h[2p+1] - h[2p]   u3 = h[2p] - h[2p+2]        (fp32, then the 2-piece split)
//   V0 = g0   V1 = (g0 + g1 + g2) / 2   V2 = (g0 - g1 + g2) / 2   V3 = -g2                              (fd_commit_weights)
//   m_j = V_j . u_j        y[2p] = bias + m0 + m1 + m2        y[2p+1] = bias + m1 - m2 + m3
// Every m_j is a 2-piece product with fp32 accumulation like the direct form's; the sums of three of them add three fp32 roundings.
// Image: [block][entry][pair p][j][piece][64 ch] fp16 = 1 KB per pair, 16-byte slot s of a 256-byte sub-row stored at s ^ (p & 15).
// A work item = (block, 128-column group, entry, 32 pairs = 64 frames): its 32 KB window is one linear DMA; 48 matrix instructions and
// 32 stores per wave.  Registers: 128 of stationary weight pieces, acc + lo, and the two output tiles as finished fp32 values.
// =================================================================================================
constexpr int GW_PAIRS = 32;                      // pairs per item (64 frames)
constexpr int GW_ROWB = 1024;                     // bytes per pair row: [4 j][2 pieces][64 ch] fp16
constexpr int GW_WINB = GW_PAIRS * GW_ROWB;       // 32 KB per window = 8 DMA rounds of 4 KB
__host__ __device__ inline int gw_pairs(int T) { return ((T + 2 * GW_PAIRS - 1) / (2 * GW_PAIRS)) * GW_PAIRS; }      // image rows (pairs) per (block, entry)

// h (fp32 [3][B][64][T]) -> the transformed piece image.  Thread = (pair p, 8-channel group g): eight neighbouring lanes write the eight
// 16-byte slots of one 128-byte half of a sub-row (whole lines), and read their channels' four frames 2p-1 .. 2p+2 (the inner two as one
// 8-byte load); frames outside the utterance are zero, pairs behind it are written as zeros.
__global__ void __launch_bounds__(256) k_h_wino(const float *__restrict__ h, char *__restrict__ wx, int *__restrict__ range_flag, int B, int T,
                                                int P, const int *__restrict__ lens)
{
    const int bb = blockIdx.y;                           // blk*B + b
    const int e = blockIdx.x * 256 + threadIdx.x, p = e >> 3, g = e & 7;
    if (p >= P) return;
    if (range_flag[32] != 0) {      // h did not fit in the previous step: the fp32 GEMM takes this one as well
        if (e == 0 && bb == 0) atomicOr(range_flag, 1);
        return;
    }
    const int Tb = frames_of(lens, bb % B, T);
    const float *hb = h + ((int64_t)bb * fd::HID + 8 * g) * T;
    const int t1 = 2 * p;                                // frames t1 - 1, t1, t1 + 1, t1 + 2
    float d[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float *hc = hb + (int64_t)c * T;
        d[c][0] = (t1 - 1 >= 0 && t1 - 1 < Tb) ? hc[t1 - 1] : 0.0f;
        d[c][1] = (t1 < Tb) ? hc[t1] : 0.0f;
        d[c][2] = (t1 + 1 < Tb) ? hc[t1 + 1] : 0.0f;
        d[c][3] = (t1 + 2 < Tb) ? hc[t1 + 2] : 0.0f;
    }
    float mx = 0.0f;
    char *row = wx + ((int64_t)bb * P + p) * GW_ROWB;
    const int sw = p & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float u[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            u[c] = j == 0 ? d[c][0] - d[c][2] : (j == 1 ? d[c][1] + d[c][2] : (j == 2 ? d[c][2] - d[c][1] : d[c][1] - d[c][3]));
            mx = fmaxf(mx, fabsf(u[c]));
        }
        float4 ph, pl;
        split8(u, ph, pl);
        *reinterpret_cast<float4 *>(row + j * 256 + ((g ^ sw) << 4)) = ph;
        *reinterpret_cast<float4 *>(row + j * 256 + (((g + 8) ^ sw) << 4)) = pl;
    }
    if (!(mx < GX_LIMIT)) atomicOr(range_flag, 1);      // also catches NaN / inf
}

__device__ __forceinline__ void gw_dma(const char *wx, char *lds_buf, const GxItem &it, int B, int P, int wave_u, int lane)
{
#ifdef FD_GW_NO_FETCH      // probe: what do the window copies cost the short items?
    return;
#endif
    const char *src = wx + (((int64_t)it.blk * B + it.b) * P + it.chunk * GW_PAIRS) * GW_ROWB + wave_u * 1024;    // uniform
    const unsigned dst = (unsigned)(uintptr_t)(lds_ptr_t)(lds_buf + wave_u * 1024);
    const unsigned voff = lane * 16;
    unsigned keep;
#pragma unroll
    for (int j = 0; j < GW_WINB / 4096; ++j)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(src + j * 4096), "s"(dst + j * 4096)
                     : "memory");
}

// One item: 32 pairs.  Vector-memory order per item: [DMA of the next window] [16 stores of the even frames] [16 of the odd ones].
// VALU instructions do not issue under the SIMD's own matrix instruction, so the combination is arranged to need as few as
// possible: ONE accumulator chain runs through m1, m2, m0 (it ends as y[2p] = m0 + m1 + m2, bias included: the bias enters once, through
// the cross-term accumulator of the first product); m1 is read out of it on the way (keep = m1 + bias), and behind m2 the chain's value
// s = m1 + m2 + bias turns keep into d = 2 keep - s = m1 - m2 + bias, which is the C operand of the m3 chain (-> y[2p+1]).
// 96 VALU instructions per item: bias 16, m1 16, d 32, and the two read-outs in front of the stores.
template <int BUF, bool FULL>
__device__ __forceinline__ void gw_item(char *lds, const GxItem &cur, bool more, const GxItem &nxt, const char *wx, float *kpack,
                                        const float4 (&wq)[2][16], float bias_lo, const int (&aoff)[2][4], int B, int T, int P,
                                        int wave_u, int lane, int Tb)
{
    const int l31 = lane & 31, hi = lane >> 5;
    if (more) gw_dma(wx, lds + (BUF ^ 1) * GW_WINB, nxt, B, P, wave_u, lane);
    const int t_begin = cur.chunk * (2 * GW_PAIRS);
    float *krow = kpack + (((int64_t)cur.blk * B + cur.b) * T + t_begin) * fd::KREC + (cur.xg * 4 + wave_u) * 32;     // uniform
    // pair row of register r: (r & 3) + 8 (r >> 2) + 4 hi -> frame 2 * that (+ 1): the lane part goes into the per-lane offset
    const unsigned loff = (unsigned)(8 * hi) * (unsigned)fd::KREC + (unsigned)l31;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(krow, 0, 2 * GW_PAIRS * fd::KREC * 4, 0x00020000);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const char *hb = lds + BUF * GW_WINB;
    constexpr int ORDER[4] = {1, 2, 0, 3};
    f32x16 keep, acc, lo;
#pragma unroll
    for (int r = 0; r < 16; ++r) lo[r] = bias_lo;      // 2048 * bias: it travels through the chain into both outputs
    float4 a1, a2;
    auto store16 = [&](int odd) {
#ifdef FD_GX_NO_STORE      // probe: the matrix + combination work alone
        if (FULL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v_ = fmaf(lo[r], GX_INV_SCALE, acc[r]); asm volatile("" :: "v"(v_)); }
            (void)rs; (void)loff;
            return;
        }
#endif
        if (FULL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[r] = fmaf(lo[r], GX_INV_SCALE, acc[r]);      // (in place: no second tile of values next to the chain's registers)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), rs, loff * 4u, (2 * ((r & 3) + 8 * (r >> 2)) + odd) * fd::KREC * 4,
                                                      FD_GX_STORE_AUX);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t_begin + 2 * drow(r, hi) + odd < Tb) (krow + (2 * ((r & 3) + 8 * (r >> 2)) + odd) * fd::KREC)[loff] = fmaf(lo[r], GX_INV_SCALE, acc[r]);
        }
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int j = ORDER[s];
        // (operands are requested one k step ahead inside a product, not across products: the registers that would hold them over the
        // read-outs between two products are the ones this kernel does not have; the CU's other workgroup covers the four exposed reads)
        a1 = *reinterpret_cast<const float4 *>(hb + aoff[0][0] + j * 256);
        a2 = *reinterpret_cast<const float4 *>(hb + aoff[1][0] + j * 256);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int kg = j * 4 + k4;
            float4 n1 = a1, n2 = a2;
            if (k4 + 1 < 4) {
                n1 = *reinterpret_cast<const float4 *>(hb + aoff[0][k4 + 1] + j * 256);
                n2 = *reinterpret_cast<const float4 *>(hb + aoff[1][k4 + 1] + j * 256);
            }
            // the chain starts from 0 (m1) and, for m3, from d; its cross-term accumulator from 2048 * bias (m1) and from 0 (m3)
            acc = mfma_f16(a1, wq[0][kg], (k4 == 0 && s == 0) ? zero : ((k4 == 0 && s == 3) ? keep : acc));
            lo = mfma_f16(a2, wq[0][kg], (k4 == 0 && s == 3) ? zero : lo);
            lo = mfma_f16(a1, wq[1][kg], lo);
            a1 = n1;
            a2 = n2;
        }
        if (s == 0) {                // keep = m1 + bias
#pragma unroll
            for (int r = 0; r < 16; ++r) keep[r] = fmaf(lo[r], GX_INV_SCALE, acc[r]);
        } else if (s == 1) {         // the chain holds s = m1 + m2 + bias: keep = d = 2 (m1 + bias) - s = m1 - m2 + bias
#pragma unroll
            for (int r = 0; r < 16; ++r) keep[r] = fmaf(2.0f, keep[r], -fmaf(lo[r], GX_INV_SCALE, acc[r]));
        } else store16(s == 3 ? 1 : 0);      // y[2p] = s + m0, then y[2p+1] = d + m3
    }
    if (more) {      // the DMA has landed; this item's 32 buffer stores may still fly
#ifdef FD_GX_NO_STORE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (FULL) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
    __builtin_amdgcn_s_barrier();
}

__global__ void __launch_bounds__(256, 2) k_kp_gemm_w(const char *__restrict__ wx /*[3][B][P][4][2][64] fp16*/, float *__restrict__ kpack,
                                                      const float4 *g0, const float4 *g1, const float4 *g2, const float *gb0,
                                                      const float *gb1, const float *gb2, const int *__restrict__ range_flag, int B,
                                                      int T, int P, int chunks_per_utt, int n_items, const int *__restrict__ lens)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * GW_WINB];     // 2 x 32 KB
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int XG = fd::KREC / 128;
    if (*range_flag != 0) return;      // out-of-range operands: the fp32 kernel behind us does this step
    // the schedule of k_kp_gemm_h2: whole column groups in step first (a window is fetched from HBM once per XCD), equal id ranges last
    const int ny = B * chunks_per_utt, n_wg = gridDim.x, w = blockIdx.x;
    const int q = (fd::NBLK * XG) / n_wg, base = q * n_wg * ny, rest = n_items - base;
    const int r0 = (int)((int64_t)w * rest / n_wg), r1 = (int)((int64_t)(w + 1) * rest / n_wg);
    const int n_mine = q * ny + (r1 - r0);
    if (n_mine <= 0) return;
    auto decode = [&](int id) {
        GxItem it;
        it.blk = id / (XG * ny);
        const int rem = id - it.blk * (XG * ny);
        it.xg = rem / ny;
        const int yy = rem - it.xg * ny;
        it.b = yy / chunks_per_utt;
        it.chunk = yy - it.b * chunks_per_utt;
        return it;
    };
    auto advance = [&](GxItem it) {
        if (++it.chunk == chunks_per_utt) {
            it.chunk = 0;
            if (++it.b == B) {
                it.b = 0;
                if (++it.xg == XG) { it.xg = 0; ++it.blk; }
            }
        }
        return it;
    };
    int run = 0, left = (q > 0) ? ny : (r1 - r0);
    GxItem cur = decode((q > 0) ? w * ny : base + r0);
    // byte offsets of the A-operand reads: pair row l31, slot = (8*piece + 2*k4 + hi) ^ (row & 15); the sub-row j adds the constant 256 j
    int aoff[2][4];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) aoff[q2][k4] = l31 * GW_ROWB + (((q2 * 8 + k4 * 2 + hi) ^ (l31 & 15)) << 4);
    gw_dma(wx, lds, cur, B, P, wave_u, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float4 wq[2][16];
    float bias_lo = 0.0f;      // 2048 * bias of this lane's column (one register: gw_item copies it into the accumulator twice per item)
    int have_blk = -1, have_xg = -1;
#pragma unroll 1
    for (int i = 0; i < n_mine; i += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && i + 1 >= n_mine) break;
            if (cur.blk != have_blk || cur.xg != have_xg) {
                const float4 *gp = cur.blk == 0 ? g0 : (cur.blk == 1 ? g1 : g2);
                const float *gb = cur.blk == 0 ? gb0 : (cur.blk == 1 ? gb1 : gb2);
                const int ptile = cur.xg * 4 + wave_u;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int kg = 0; kg < 16; ++kg) wq[q2][kg] = gp[(((int64_t)ptile * 2 + q2) * 16 + kg) * 64 + lane];
                bias_lo = gb[ptile * 32 + l31] * GX_SCALE;
                have_blk = cur.blk; have_xg = cur.xg;
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            const bool more = (i + half + 1 < n_mine);
            GxItem nxt = cur;
            if (more) {
                if (left > 1) nxt = advance(cur);
                else {
                    ++run;
                    nxt = decode(run < q ? (run * n_wg + w) * ny : base + r0);
                    left = (run < q ? ny : r1 - r0) + 1;
                }
            }
            --left;
            const int Tb = frames_of(lens, cur.b, T);
            const bool full = cur.chunk * (2 * GW_PAIRS) + 2 * GW_PAIRS <= Tb;
            if (half == 0) {
                if (full) gw_item<0, true>(lds, cur, more, nxt, wx, kpack, wq, bias_lo, aoff, B, T, P, wave_u, lane, Tb);
                else gw_item<0, false>(lds, cur, more, nxt, wx, kpack, wq, bias_lo, aoff, B, T, P, wave_u, lane, Tb);
            } else {
                if (full) gw_item<1, true>(lds, cur, more, nxt, wx, kpack, wq, bias_lo, aoff, B, T, P, wave_u, lane, Tb);
                else gw_item<1, false>(lds, cur, more, nxt, wx, kpack, wq, bias_lo, aoff, B, T, P, wave_u, lane, Tb);
            }
            cur = nxt;
        }
    }
}

}  // namespace fdk_fast

// ------------------------------------------------------------------------------------------------
// stage drivers
// ------------------------------------------------------------------------------------------------
namespace fdk {
using namespace fdk_fast;

hipError_t fast_kp_front(const Launch &L, const StepIO &io, int B, int T)
{
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    const dim3 grid((T + KPF_VALID - 1) / KPF_VALID, B, fd::NBLK);
    const int *run_if = nullptr;
    const char *name = "kp_front";
    c->h_image_ready = false;
    const Pipe pipe = fd_pipe(c, c->conv_f16 && w.kpf_f16_ok, 19);
    if (pipe != PIPE_F32_ONLY) {
        KpFrontW2 k2;
        for (int n = 0; n < fd::NBLK; ++n) {
            k2.in_pack[n] = reinterpret_cast<const float4 *>(w.kp_in_h2[n]); k2.in_b[n] = w.blk[n].kp_in.b;
            for (int l = 0; l < 6; ++l) { k2.res_pack[n][l] = reinterpret_cast<const float4 *>(w.kp_res_h2[n][l]); k2.res_b[n][l] = w.blk[n].kp_res[l].b; }
        }
        // the direct fp16x2 GEMM reads the piece image this kernel can write on its way out; the Winograd form builds its own (k_h_wino)
        // and the fp32 GEMM reads hout
        const bool image = fd_pipe(c, c->gemm_f16 && w.gemm_f16_ok, 0) != PIPE_F32_ONLY && !(c->gemm_wino && w.gemm_w_ok);
        FD_LAUNCH(L, name, k_kp_front_h2, grid, dim3(256), 0, io.mel, c->ws.kp_hB, image ? reinterpret_cast<char *>(c->ws.h_f16) : (char *)nullptr, k2,
                  (const float *)c->ws.noise, (const StepParams *)c->ws.params, io.sampler, B, T, gx_rows(T), c->ws.range_flag, c->step_lens);
        c->h_image_ready = image;      // the GEMM's fp16 image of h is written (k_h_split not needed)
        run_if = c->ws.range_flag + 19;
        name = "kp_front_fp32_fallback";
        if (pipe == PIPE_F16_ONLY) return hipSuccess;
    }
    KpFrontW kw;
    for (int n = 0; n < fd::NBLK; ++n) {
        kw.in_pack[n] = w.kp_in_pack[n]; kw.in_b[n] = w.blk[n].kp_in.b;
        for (int l = 0; l < 6; ++l) { kw.res_pack[n][l] = w.kp_res_pack[n][l]; kw.res_b[n][l] = w.blk[n].kp_res[l].b; }
    }
    FD_LAUNCH(L, name, k_kp_front, grid, dim3(256), 0, io.mel, c->ws.kp_hB, kw, (const float *)c->ws.noise,
              (const StepParams *)c->ws.params, io.sampler, B, T, run_if, c->step_lens);
    return hipSuccess;
}

hipError_t fast_kp_gemm(const Launch &L, int B, int T)
{
    // all three blocks in one persistent launch, 2 workgroups per CU (a launch per block, next to or between the LVC layers, was
    // measured and did not pay: LABBOOK.md)
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    const int tiles_per_utt = (T + 31) / 32;
    const int chunks_per_utt = (tiles_per_utt + GEMM_CT - 1) / GEMM_CT;
    const int chunk_tiles = (tiles_per_utt + chunks_per_utt - 1) / chunks_per_utt;     // balanced, <= GEMM_CT
    const int n_items = fd::NBLK * (fd::KREC / 128) * B * chunks_per_utt;
    const int grid = n_items < 2 * c->num_cus ? n_items : 2 * c->num_cus;              // persistent: 2 workgroups per CU
    const Pipe pipe = fd_pipe(c, c->gemm_f16 && w.gemm_f16_ok, 0);
    const bool f16 = pipe != PIPE_F32_ONLY;
    if (f16 && c->gemm_wino && w.gemm_w_ok) {
        // Winograd F(2,3) over the frame axis: the transformed piece image from the front's fp32 h, then 2/3 of the direct form's matrix work
        const int P = gw_pairs(T), chunks = P / GW_PAIRS, items = fd::NBLK * (fd::KREC / 128) * B * chunks;
        const int grid2 = items < 2 * c->num_cus ? items : 2 * c->num_cus;
        FD_LAUNCH(L, "h_wino", k_h_wino, dim3((8 * P + 255) / 256, fd::NBLK * B), dim3(256), 0, (const float *)c->ws.kp_hB,
                  reinterpret_cast<char *>(c->ws.h_f16), c->ws.range_flag, B, T, P, c->step_lens);
        FD_LAUNCH(L, "kp_gemm_f16x2", k_kp_gemm_w, dim3(grid2), dim3(256), 0, reinterpret_cast<const char *>(c->ws.h_f16), c->ws.kpack,
                  reinterpret_cast<const float4 *>(w.gemm_w_pack[0]), reinterpret_cast<const float4 *>(w.gemm_w_pack[1]),
                  reinterpret_cast<const float4 *>(w.gemm_w_pack[2]), w.gemm_bias[0], w.gemm_bias[1], w.gemm_bias[2],
                  (const int *)c->ws.range_flag, B, T, P, chunks, items, c->step_lens);
        if (pipe == PIPE_F16_ONLY) return hipSuccess;
    } else if (f16) {
        const int R = gx_rows(T);
        const int chunks = (T + GX_CT * 32 - 1) / (GX_CT * 32), items = fd::NBLK * (fd::KREC / 128) * B * chunks;
        const int grid2 = items < 2 * c->num_cus ? items : 2 * c->num_cus;
        if (!c->h_image_ready)      // the fp16-pipe predictor front writes the image itself
            FD_LAUNCH(L, "h_split", k_h_split, dim3((32 * R + 255) / 256, fd::NBLK * B), dim3(256), 0, (const float *)c->ws.kp_hB,
                      reinterpret_cast<unsigned *>(c->ws.h_f16), c->ws.range_flag, B, T, R, c->step_lens);
        FD_LAUNCH(L, "kp_gemm_f16x2", k_kp_gemm_h2, dim3(grid2), dim3(256), 0, reinterpret_cast<const char *>(c->ws.h_f16), c->ws.kpack,
                  reinterpret_cast<const float4 *>(w.gemm_h2_pack[0]), reinterpret_cast<const float4 *>(w.gemm_h2_pack[1]),
                  reinterpret_cast<const float4 *>(w.gemm_h2_pack[2]), w.gemm_bias[0], w.gemm_bias[1], w.gemm_bias[2],
                  (const int *)c->ws.range_flag, B, T, R, chunks, items, c->step_lens, 0, fd::NBLK);
        if (pipe == PIPE_F16_ONLY) return hipSuccess;
    }
    // fp32 matrix pipe: the whole job when the fp16 form is off, otherwise an early-exit launch that only works when
    // k_h_split found operands outside the fp16 range
    FD_LAUNCH(L, f16 ? "kp_gemm_fp32_fallback" : "kp_gemm", k_kp_gemm, dim3(grid), dim3(256), 0, (const float *)c->ws.kp_hB, c->ws.kpack,
              w.gemm_pack[0], w.gemm_pack[1], w.gemm_pack[2], w.gemm_bias[0], w.gemm_bias[1], w.gemm_bias[2], B, T, chunks_per_utt,
              chunk_tiles, n_items, f16 ? (const int *)c->ws.range_flag : (const int *)nullptr, c->step_lens);
    return hipSuccess;
}

}  // namespace fdk
