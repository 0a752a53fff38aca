// fd_kernels_fast.hip -- the gfx950 kernel set of the FastDiff denoiser step.
//
// Everything with a channel contraction runs on the exact-fp32 matrix pipe, v_mfma_f32_32x32x2_f32
// (bit-for-bit an fp32 fmaf chain, so the fp32 parity bar of the reference holds):
//   D[row][col] += A[row][k] * B[k][col],   wave64 operand layout (MI355X_MICROARCH / cdna_hip guide 3):
//     A: lane l holds A[row = l&31][k = l>>5]          (1 VGPR)
//     B: lane l holds B[k = l>>5][col = l&31]          (1 VGPR)
//     D: lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]   (16 VGPRs)
// Convolutions map as: row = output channel, col = time, k = (tap, input channel); the B operand is read
// straight out of an LDS-staged sliding window (32 consecutive time samples per half-wave: conflict free),
// the A operand (weights, or the predicted per-frame kernel of the location-variable convolution) sits in
// registers, pre-packed in HBM as [s4 = step/4][lane][4] so each lane fetches 4 k-steps with one 16 B load.
#include "fd_kernels.h"
#include "fd_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace fdk_fast {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int drow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
// On gfx950 fp32 VALU instructions do NOT execute under a running fp32 MFMA (tools/ubench/coexec_probe.hip: every extra
// VALU op costs ~4 cycles of matrix-pipe time, also when it comes from the other wave of the SIMD), so the inner loops
// must not spend VALU on addressing.  The compiler likes to pair LDS reads of neighbouring taps into ds_read2_b32, whose
// 8-bit offsets then force a fresh v_add per row.  Hiding the relation between the per-tap offsets keeps every read a
// plain ds_read_b32 with a 16-bit immediate from three fixed base registers.
__device__ __forceinline__ int opaque(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float lrelu(float v, float s)   // 0 < s < 1:  max(v, s*v)
{
    // median(v, s*v, +inf) == max(v, s*v).  One v_med3_f32 instead of fmaxf()'s canonicalise + v_max pair, and -- unlike
    // an inline-asm v_max -- visible to the compiler's VALU->MFMA hazard padding.
    return __builtin_amdgcn_fmed3f(v, v * s, __builtin_inff());
}
__device__ __forceinline__ float f4c(const float4 &v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }

// Ragged batches: `lens` (nullable, device) holds the valid frames of every utterance of a zero-padded batch.  Every kernel
// then treats utterance b as if it were lens[b] frames long: positions behind it read as the zero padding a convolution sees
// at the end of a signal, tiles behind it are skipped.  Inside [0, lens[b]) the result is bit-identical to running the
// utterance alone (tests/test_gpu_parity.py); what the output buffers hold behind it is unspecified.
__device__ __forceinline__ int frames_of(const int *lens, int b, int T) { return lens ? lens[b] : T; }

// Range flags (Workspace::range_flag): word i = "an operand of fp16-pipe launch i did not fit in this step", word 32 + i = the
// same for the previous step of the sampler (copied by k_advance).  A launch whose flag was raised in the previous step does
// not try again: it raises its flag at once and leaves the step to the fp32 kernel behind it -- a trajectory that has left the
// fp16 range (an untrained network over 1000 steps does) then costs the fp32 kernels only, not both.
__device__ __forceinline__ bool skip_after_previous_overflow(int *flag)
{
    if (flag[32] == 0) return false;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) atomicOr(flag, 1);
    return true;
}

// ---- 2-piece fp16 operands: v = v1 + 2^-11 v2, v1 = fp16(v), v2 = fp16((v - v1) * 2^11) ----------------
constexpr float GX_SCALE = 2048.0f, GX_INV_SCALE = 1.0f / 2048.0f;
constexpr float GX_LIMIT = 32768.0f;            // magnitudes from here on do not fit: the kernels raise a range flag
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f16(const float4 &a, const float4 &b, f32x16 c)
{
    union { float4 f; f16x8 h; } ua, ub;
    ua.f = a;
    ub.f = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.h, ub.h, c, 0, 0, 0);
}

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo)
{
    const f2_t v = {a, b};
    const h2_t h = __builtin_convertvector(v, h2_t);                       // v_cvt_pk_f16_f32, round to nearest even
    const f2_t r = (v - __builtin_convertvector(h, f2_t)) * GX_SCALE;       // exact
    const h2_t l = __builtin_convertvector(r, h2_t);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split8(const float (&v)[8], float4 &hi, float4 &lo)
{
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2(v[2 * i], v[2 * i + 1], h[i], l[i]);
    hi = make_float4(__uint_as_float(h[0]), __uint_as_float(h[1]), __uint_as_float(h[2]), __uint_as_float(h[3]));
    lo = make_float4(__uint_as_float(l[0]), __uint_as_float(l[1]), __uint_as_float(l[2]), __uint_as_float(l[3]));
}
__device__ __forceinline__ float amax4(float m, const float4 &v) { return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); }
// byte offset of 16 B slot `slot` (piece*4 + channel/8) of row `row` in a [row][128 B] piece image
__device__ __forceinline__ int h2_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }
// 8 channels (image slot `slot`, 0..3) of one conv tap at image row `row`: the products the matrix pipe forms for the other
// columns -- a1 += w1.x1, a2 += w1.x2 + w2.x1, fp32 accumulation -- on v_dot2_f32_f16.  w1/w2 = the weight pieces of those channels.
__device__ __forceinline__ void halo_dot(const char *xs, int row, int slot, const float4 &w1f, const float4 &w2f, float &a1, float &a2)
{
    union { float4 f; h2_t h[4]; } x1, x2, w1, w2;
    x1.f = *reinterpret_cast<const float4 *>(xs + h2_off(row, slot));
    x2.f = *reinterpret_cast<const float4 *>(xs + h2_off(row, 4 + slot));
    w1.f = w1f;
    w2.f = w2f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a1 = __builtin_amdgcn_fdot2(w1.h[q], x1.h[q], a1, false);
        a2 = __builtin_amdgcn_fdot2(w1.h[q], x2.h[q], a2, false);
        a2 = __builtin_amdgcn_fdot2(w2.h[q], x1.h[q], a2, false);
    }
}


// Cache policy of the big streams.  FD_LVC_NT bits: which accesses carry the nt (non-temporal) bit -- hop-64/256 LVC layers: 1 = x
// loads, 2 = out stores, 4 = the frame's record, 8 = skip loads; 16 = hop-8 LVC out stores, 32 = ConvTranspose out stores, 64 =
// first_conv out stores.  Measured (profiles/r03/s40_s41_nt_policy.txt): 2 pays (layer -3 % alone, call -0.75 % at B=8, -1.9 % at B=1),
// 4 costs 6 % (both waves of a row tile read the record: it has to stay in L1/L2), the rest +-0.
#ifndef FD_LVC_NT
#define FD_LVC_NT 2
#endif
typedef float lvc_f4 __attribute__((ext_vector_type(4)));
template <int BIT>
__device__ __forceinline__ float4 lvc_ld(const float4 *p)
{
    if constexpr ((FD_LVC_NT & BIT) != 0) {
        const lvc_f4 v = __builtin_nontemporal_load(reinterpret_cast<const lvc_f4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else return *p;
}
template <int BIT>
__device__ __forceinline__ void lvc_st(float *p, float v) { if constexpr ((FD_LVC_NT & BIT) != 0) __builtin_nontemporal_store(v, p); else *p = v; }
template <int BIT>
__device__ __forceinline__ void lvc_st(float4 *p, const float4 &v)
{
    if constexpr ((FD_LVC_NT & BIT) != 0) __builtin_nontemporal_store(lvc_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<lvc_f4 *>(p));
    else *p = v;
}

// =================================================================================================
// a3: first_audio_conv  Conv1d(1,32,k7,pad3)  (FastDiff_model.py:34-36,89)   -- VALU, HBM-write bound
// =================================================================================================
// advance != null (a sampler step that is not the first of its graph / launch sequence): the first workgroup does the previous step's
// end-of-step bookkeeping on the way -- next row of the step table, that step's range flags become "previous step" and join the call's
// sticky set -- what k_advance does in a launch of its own.  Nothing else in this kernel reads either; the kernels behind it start
// after the whole grid.
// V (option "first_variant"; a probe of round 4's two-process bisect, profiles/r04/s8_*: with this kernel replaced by the naive one the
// library's sampler is no longer disturbed by short-lived neighbour processes): 0 = as built since round 1 -- the weights arrive by
// scalar loads (uniform index), x and a0 through the vector L1; 1 = the weights through vector loads + LDS (no scalar data load in the
// kernel); 2 = x read at system scope (sc0 sc1: past the vector L1); 3 = both; 4 = the scalar-load form behind a dummy LDS write +
// barrier (the timing of form 1 without its loads); 5 = the weights through vector loads straight from global memory (no LDS, no barrier).
template <int V>
__global__ void __launch_bounds__(256) k_first_conv(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ bias, float *__restrict__ a0, int L,
                                                    const int *__restrict__ lens, StepParams *advance, int *__restrict__ range_flags)
{
    if (advance && blockIdx.x == 0 && blockIdx.y == 0) {
        if (threadIdx.x == 0) advance->step_idx += 1;
        if (threadIdx.x < 32) {
            const int f = range_flags[threadIdx.x];
            range_flags[32 + threadIdx.x] = f;
            range_flags[64 + threadIdx.x] |= f;
            range_flags[threadIdx.x] = 0;
        }
    }
    constexpr bool WLDS = (V == 1 || V == 3), WVEC = (V == 5);
    __shared__ float wl[(WLDS || V == 4) ? fd::C * 8 : 1];      // [out][7 taps + bias]
    if constexpr (WLDS) {
        const int o = threadIdx.x >> 3, k = threadIdx.x & 7;
        wl[threadIdx.x] = k < 7 ? w[o * 7 + k] : bias[o];
        __syncthreads();
    }
    if constexpr (V == 4) {      // probe: the scalar-load form with the other form's LDS write + barrier in front (timing only)
        wl[threadIdx.x] = (float)threadIdx.x;
        __syncthreads();
        if (wl[(threadIdx.x + 1) & 255] < 0.0f) return;
    }
    int vz = 0;
    if constexpr (WVEC) asm volatile("v_mov_b32 %0, 0" : "=v"(vz));      // probe: an index the compiler cannot prove uniform -> vector loads, no LDS
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int Lb = lens ? lens[b] * fd::HOPT : L;          // this utterance's own length (ragged batch)
    if (t0 >= Lb) return;
    const float *xr = x + (int64_t)b * L;
    float xv[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int p = t0 - 3 + i;
        if constexpr (V == 2 || V == 3) xv[i] = (p >= 0 && p < Lb) ? __hip_atomic_load(xr + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0f;
        else xv[i] = (p >= 0 && p < Lb) ? xr[p] : 0.0f;
    }
#pragma unroll 4
    for (int o = 0; o < fd::C; ++o) {
        float wv[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) wv[k] = WLDS ? wl[o * 8 + k] : w[o * 7 + k + vz];
        const float bv = WLDS ? wl[o * 8 + 7] : bias[o + vz];
        float4 r = make_float4(bv, bv, bv, bv);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            r.x += wv[k] * xv[k]; r.y += wv[k] * xv[k + 1]; r.z += wv[k] * xv[k + 2]; r.w += wv[k] * xv[k + 3];
        }
        lvc_st<64>(reinterpret_cast<float4 *>(a0 + ((int64_t)b * fd::C + o) * L + t0), r);
    }
}

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
    // scalar DATA loads are what a short-lived neighbour process on the same compute units can disturb (k_first_conv)
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
                uint2 ph, pl;
                split2(v[0], v[1], ph.x, pl.x);
                split2(v[2], v[3], ph.y, pl.y);
                *reinterpret_cast<uint2 *>(irow + (((mt * 4 + j) ^ sw) << 4) + 8 * hi) = ph;
                *reinterpret_cast<uint2 *>(irow + (((8 + mt * 4 + j) ^ sw) << 4) + 8 * hi) = pl;
            }
        }
    }
    // the image's padding rows (0 and T+1 .. R-1) must read as zeros: first and last tile of the utterance write them
    {
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
#ifndef FD_GX_STORE_AUX
#define FD_GX_STORE_AUX 0      // cache policy bits of the predicted-kernel stores (2 = nt)
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
            lo = mfma_f16(a1, wq[1][kg], lo);
            a1 = n1;
            a2 = n2;
        }
        GX_STAMP(2);
#ifdef FD_GX_NO_STORE
        if (acc[0] != 12345.678f) continue;
#endif
        if (FULL) {
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

// =================================================================================================
// a7+a8+a9: one whole LVC layer (modules.py:208-217), fused:
//   x' = x + skip ; y = lrelu(conv_{k3,dil}(lrelu(x'))) ; z = LVC(y; K_f, bias_f) ; out = x' + sigmoid(z[:32])*tanh(z[32:])
// HBM traffic is the algorithmic minimum of the layer: read x, skip, the frame's predicted kernel; write out.
// Workgroup = 4 waves x WC columns.  Each wave: dilated conv on its WC columns + the 2 halo columns the
// LVC taps need (one extra MFMA tile), y kept wave-private in LDS, then
//   HOP >= 64: LVC on the matrix pipe, A = the frame's 64x96 predicted kernel (96 VGPRs), 2 row tiles share B
//   (hop 8 has its own all-VALU kernel, k_lvc_h8)
// =================================================================================================
template <int HOP, int DIL>
struct LvcCfg {
    static constexpr int WC = 64;                               // columns per wave
    static constexpr int W = 4 * WC;                            // columns per workgroup
    static constexpr int H = (DIL + 1 + 3) & ~3;                // staged halo (multiple of 4 for 16 B loads)
    static constexpr int XLD = W + 2 * H;
    static constexpr int YLD = W + 4;                           // y columns -1 .. W, padded to a multiple of 4
};

#ifdef FD_LVC_TIMING
__device__ long long fd_dbg[64 * 4 * 8];
#define FD_STAMP(i) do { if (lane == 0 && blockIdx.y == FD_LVC_TIMING && blockIdx.x >= 100 && blockIdx.x < 164) \
        fd_dbg[((blockIdx.x - 100) * 4 + wave) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#elif defined(FD_LVC_TIMELINE)
// every workgroup's wave 0: s_memrealtime at each phase boundary + where it ran (tools/ubench/lvc_h2_timeline.hip)
__device__ long long *fd_tl;
#define FD_STAMP(i) do { if (threadIdx.x == 0) { long long *q_ = fd_tl + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 10; \
        q_[i] = (long long)__builtin_amdgcn_s_memrealtime(); \
        if ((i) == 0) { q_[8] = __builtin_amdgcn_s_getreg(63492); q_[9] = __builtin_amdgcn_s_getreg(63508); } } } while (0)
#else
#define FD_STAMP(i)
#endif

// sigmoid(a) * tanh(b) with two exponentials and one reciprocal:  (1 - v) / ((1 + u)(1 + v)),  u = e^-a, v = e^-2b.
// Only v needs a guard: v = inf (b << 0) would give inf/inf; u = inf or 0 and v = 0 are the correct saturated results.
__device__ __forceinline__ float gate(float a, float b)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    b = fmaxf(b, -15.0f);
    const f2 e = f2{a, b} * f2{-1.4426950408889634f, -2.8853900817779268f};
    const f2 uv = f2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    const f2 d = uv + 1.0f;
    return (1.0f - uv.y) * __builtin_amdgcn_rcpf(d.x * d.y);
}

template <int HOP, int DIL>
__global__ void __launch_bounds__(256, 2) k_lvc_layer(const float *__restrict__ xin, const float *__restrict__ skip,
                                                      float *__restrict__ xout, const float *__restrict__ kpack, int layer,
                                                      const float *__restrict__ wpack, const float *__restrict__ wref,
                                                      const float *__restrict__ cbias, int T, const int *__restrict__ run_if,
                                                      const int *__restrict__ lens)
{
    static_assert(HOP >= 64, "hop 8 has its own kernel (k_lvc_h8)");
    using Cfg = LvcCfg<HOP, DIL>;
    if (run_if && *run_if == 0) return;      // fallback launch behind k_lvc_h2: only when that kernel flagged its operands
    constexpr int WC = Cfg::WC, W = Cfg::W, H = Cfg::H, XLD = Cfg::XLD, YLD = Cfg::YLD, NT = WC / 32;
    // LVC work split (hop >= 64).  hop 256: the whole tile is ONE frame, so the waves split the 64 output rows instead of
    // re-loading the same kernel four times: wave = (row tile mt, column half), 4 column tiles each, 48 operand registers.
    // hop 64: a wave owns one frame (64 columns) and both row tiles.
    constexpr int LT = (HOP == 256) ? 1 : 2;           // row tiles per wave
    constexpr int LN = (HOP == 256) ? 4 : 2;           // column tiles per wave
    constexpr bool PREACT = (HOP == 256);              // xs holds leaky_relu(x'), the raw residual lives in registers (hop 64: no register room)
    __shared__ __attribute__((aligned(16))) float xs[fd::C * XLD];
    __shared__ __attribute__((aligned(16))) float ys[fd::C * YLD];
    const int Ln = T * HOP;                         // row stride of the activations
    const int b = blockIdx.y, w0 = blockIdx.x * W;
    const int Tb = frames_of(lens, b, T), Lnb = Tb * HOP;      // this utterance's own length (ragged batch): every bound below
    if (w0 >= Lnb) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int cw = wave * WC;                       // first conv column of this wave inside the tile
    const bool wave_valid = (w0 + cw) < Lnb;         // hop>=64: a wave owns whole frames; hop 8: checked per frame below
    const int mt0 = (HOP == 256) ? (wave & 1) : 0;
    const int lcw = (HOP == 256) ? 128 * (wave >> 1) : cw;     // first LVC column of this wave
    FD_STAMP(0);

    // ---- every global read is issued up front in the order of its latency; the first wait is at the first use --------------
    // (a) the frame's predicted kernel + bias: A operand of the LVC (HBM)
    float4 ka[LT][12];
    float4 bz[LT][4];
    if constexpr (HOP >= 64) {
        if (wave_valid) {
            const int f = (w0 + lcw) / HOP;
            const float *rec = kpack + ((int64_t)b * T + f) * fd::KREC;
            const float4 *kp4 = reinterpret_cast<const float4 *>(rec + layer * fd::KLAYER) + 2 * lane;   // [mt][kg][lane][8 k]
            // D rows of a lane are {0..3, 8..11, 16..19, 24..27} + 4*hi: four 16 B loads per 32-row tile
            const float4 *kb4 = reinterpret_cast<const float4 *>(rec + fd::KW + layer * 64);
#pragma unroll
            for (int m = 0; m < LT; ++m) {
#pragma unroll
                for (int i = 0; i < 12; ++i) ka[m][i] = kp4[((mt0 + m) * 6 + (i >> 1)) * 128 + (i & 1)];
#pragma unroll
                for (int j = 0; j < 4; ++j) bz[m][j] = kb4[(mt0 + m) * 8 + 2 * j + hi];
            }
        }
    }
    // (b) dilated-conv weights (L2): A operand of the conv, needed right after the staging barrier
    float4 wa[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) wa[i] = reinterpret_cast<const float4 *>(wpack)[i * 64 + lane];
    float4 cb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cb[j] = reinterpret_cast<const float4 *>(cbias)[2 * j + hi];
    // (c) x and skip tiles with halo (HBM), all in flight before the first LDS write
    {
        const float *xr = xin + (int64_t)b * fd::C * Ln, *sr = skip + (int64_t)b * fd::C * Ln;
        constexpr int NF4 = XLD / 4, TOTAL = fd::C * NF4, NK = (TOTAL + 255) / 256;
        constexpr int NB = (HOP == 64) ? 2 : 1, KB = (NK + NB - 1) / NB;   // hop 64 holds 104 kernel registers: two batches
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            float4 xa[KB], sa[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int idx = (bt * KB + k) * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4, g = w0 - H + 4 * c4;
                const bool ok = idx < TOTAL && g >= 0 && g < Lnb;
                xa[k] = ok ? *reinterpret_cast<const float4 *>(xr + (int64_t)ci * Ln + g) : make_float4(0.f, 0.f, 0.f, 0.f);
                sa[k] = ok ? *reinterpret_cast<const float4 *>(sr + (int64_t)ci * Ln + g) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the loads above ahead of everything below
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int idx = (bt * KB + k) * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4;
                if (idx < TOTAL) {
                    const float4 r = make_float4(xa[k].x + sa[k].x, xa[k].y + sa[k].y, xa[k].z + sa[k].z, xa[k].w + sa[k].w);
                    if constexpr (PREACT) {
                        // the conv reads every element three times (once per tap): activate once here instead of per read.
                        // The raw values of the centre columns (the residual) are parked in ys until the conv overwrites it.
                        *reinterpret_cast<float4 *>(xs + ci * XLD + 4 * c4) =
                            make_float4(lrelu(r.x, 0.2f), lrelu(r.y, 0.2f), lrelu(r.z, 0.2f), lrelu(r.w, 0.2f));
                        if (c4 >= H / 4 && c4 < H / 4 + W / 4) *reinterpret_cast<float4 *>(ys + ci * YLD + 4 * c4 - H) = r;
                    } else {
                        *reinterpret_cast<float4 *>(xs + ci * XLD + 4 * c4) = r;
                    }
                }
            }
        }
    }
    // (d) halo columns: thread = (side, out channel o, quarter q of the input channels); its 24 conv weights
    //     w[o][8q..8q+7][0..2] are consecutive floats (L2), consumed after the conv
    const int hside = tid >> 7, ho = (tid & 127) >> 2, hq = tid & 3;
    __syncthreads();
    FD_STAMP(1);
    // residual values of this lane's outputs (hop >= 64): registers, so that ys can take the conv output
    float resid[PREACT ? LN : 1][PREACT ? 8 * LT : 1];
    if constexpr (PREACT) {
#pragma unroll
        for (int nt = 0; nt < LN; ++nt)
#pragma unroll
            for (int m = 0; m < LT; ++m)
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    resid[nt][m * 8 + r] = ys[(16 * (mt0 + m) + (r & 3) + 8 * (r >> 2) + 4 * hi) * YLD + lcw + nt * 32 + l31];
        __syncthreads();
    }

    // ---- dilated conv: interior columns on the matrix pipe (y index = column + 1); the LDS write-back of tile i is
    //      issued under the MFMAs of tile i+1 ---------------------------------------------------------------------------------
    if (wave_valid) {
        f32x16 acc[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = f4c(cb[r >> 2], r & 3);
            const int col = hi * XLD + H + cw + ct * 32 + l31;
            const int o[3] = {opaque(col - DIL), opaque(col), opaque(col + DIL)};
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                const int tap = s >> 4, c2 = (2 * s) & 31;
                const float xv = xs[o[tap] + c2 * XLD];
                const float v = PREACT ? xv : lrelu(xv, 0.2f);
                acc[ct] = mfma32(f4c(wa[s >> 2], s & 3), v, acc[ct]);
                if (ct > 0 && s % 3 == 1) {             // write-back of the previous tile, one row per 3 k-steps
                    const int r = s / 3, cp = cw + (ct - 1) * 32 + l31;
                    ys[drow(r, hi) * YLD + cp + 1] = (w0 + cp) < Lnb ? lrelu(acc[ct - 1][r], 0.2f) : 0.0f;
                }
            }
        }
        {
            const int cp = cw + (NT - 1) * 32 + l31;
            const bool inside = (w0 + cp) < Lnb;         // y is zero-padded for the LVC taps (modules.py:240)
#pragma unroll
            for (int r = 0; r < 16; ++r) ys[drow(r, hi) * YLD + cp + 1] = inside ? lrelu(acc[NT - 1][r], 0.2f) : 0.0f;
        }
    } else {
        // a wave past the end of the signal still owns y columns its left neighbour's taps read: they are zero padding
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) ys[drow(r, hi) * YLD + cw + ct * 32 + l31 + 1] = 0.0f;
    }
    FD_STAMP(2);
    // ---- the two halo columns (-1 and W) the LVC taps reach: 2 x 32 outputs x 96 MACs on VALU, 4 threads per output ----
    {
        float4 hwt[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) hwt[j] = reinterpret_cast<const float4 *>(wref + (ho * fd::C + 8 * hq) * 3)[j];
        const float hbias = cbias[ho];
        const int c = hside ? W : -1, g = w0 + c;
        float accv = 0.0f;
        if (g >= 0 && g < Lnb) {
            const float wv[24] = {hwt[0].x, hwt[0].y, hwt[0].z, hwt[0].w, hwt[1].x, hwt[1].y, hwt[1].z, hwt[1].w,
                                  hwt[2].x, hwt[2].y, hwt[2].z, hwt[2].w, hwt[3].x, hwt[3].y, hwt[3].z, hwt[3].w,
                                  hwt[4].x, hwt[4].y, hwt[4].z, hwt[4].w, hwt[5].x, hwt[5].y, hwt[5].z, hwt[5].w};
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int tap = 0; tap < 3; ++tap)
                {
                    const float xv = xs[(8 * hq + j) * XLD + H + c + (tap - 1) * DIL];
                    accv += wv[j * 3 + tap] * (PREACT ? xv : lrelu(xv, 0.2f));
                }
        }
        accv += __shfl_xor(accv, 1, 64);
        accv += __shfl_xor(accv, 2, 64);
        if (hq == 0) ys[ho * YLD + c + 1] = (g >= 0 && g < Lnb) ? lrelu(accv + hbias, 0.2f) : 0.0f;
    }
    FD_STAMP(3);
    __syncthreads();
    FD_STAMP(4);
    if (!wave_valid) return;

    const int64_t orow = (int64_t)b * fd::C * Ln;
    if constexpr (HOP >= 64) {
        // ---- LVC on the matrix pipe: A = rows of the frame's 64x96 predicted kernel.  With the gate-paired row order a
        //      lane holds sigmoid input (register r) and tanh input (r+8) of channel 16*mt + drow(r), r < 8.
        //      The gate/residual/store epilogue of column tile i runs under the MFMAs of tile i+1. -----------------------
        float *xo = xout + orow + (int64_t)(4 * hi) * Ln + w0 + lcw + l31;    // + channel*Ln + nt*32
        const unsigned Lnu = (unsigned)Ln;
        f32x16 a[LN][LT];
        auto epilogue_row = [&](int nt, int m, int r) {          // r < 8
            const int chl = 16 * (mt0 + m) + (r & 3) + 8 * (r >> 2);     // channel minus 4*hi
            float xr;
            if constexpr (PREACT) xr = resid[nt][m * 8 + r];
            else xr = xs[(chl + 4 * hi) * XLD + H + lcw + nt * 32 + l31];
            xo[(unsigned)chl * Lnu + (unsigned)(nt * 32)] = xr + gate(a[nt][m][r], a[nt][m][r + 8]);
        };
        constexpr int EPI = 8 * LT, GAP = 48 / EPI;       // epilogue items per column tile, k-steps between two of them
#pragma unroll
        for (int nt = 0; nt < LN; ++nt) {
#pragma unroll
            for (int m = 0; m < LT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) a[nt][m][r] = f4c(bz[m][r >> 2], r & 3);
            // k-step s <-> k = 16*(s>>3) + 8*hi + (s&7) (the record's lane order): tap = s>>4, channel = 16*((s>>3)&1) + 8*hi + (s&7)
            const int yc = hi * 8 * YLD + lcw + nt * 32 + l31;         // y index = column + 1 + (tap - 1)
            const int yo[3] = {opaque(yc), opaque(yc + 1), opaque(yc + 2)};
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                const int tap = s >> 4, c2 = 16 * ((s >> 3) & 1) + (s & 7);
                const float v = ys[yo[tap] + c2 * YLD];
#pragma unroll
                for (int m = 0; m < LT; ++m) a[nt][m] = mfma32(f4c(ka[m][s >> 2], s & 3), v, a[nt][m]);
                if (nt > 0 && s % GAP == GAP / 2) {
                    const int e = s / GAP;
                    epilogue_row(nt - 1, e / 8, e % 8);
                }
            }
            if (nt == 0) FD_STAMP(5);
        }
        FD_STAMP(6);
#pragma unroll
        for (int e = 0; e < EPI; ++e) epilogue_row(LN - 1, e / 8, e % 8);
        FD_STAMP(7);
    }
}

// =================================================================================================
// The same LVC layer on the fp16 matrix pipe (hop 64 and 256), 2-piece operands as in k_kp_gemm_h2:
//   v = v1 + 2^-11 v2 (fp16 pieces, 22 bits);  A.B ~= A1.B1 + 2^-11 (A1.B2 + A2.B1), fp32 accumulation, the cross terms in
//   their own accumulator.  Per 32x32 output tile and 96 k: 18 v_mfma_f32_32x32x16_f16 (576 cycles) instead of 48
//   v_mfma_f32_32x32x2f32 (3072 cycles).  VALU pays for the splits (3 instructions per element with v_cvt_pk_f16_f32 and
//   packed fp32 math): x' at staging, y after the conv, the predicted kernel after its load.
// LDS images are [column][piece][32 channels] fp16, 128 B per column, 16 B slot s of row r stored at s ^ ((r >> 1) & 7):
// a B operand (8 consecutive channels of one column) is one conflict-free ds_read_b128.
// Operands of magnitude >= 32768 do not fit fp16: the kernel raises *range_flag and the fp32 kernel launched behind it
// (k_lvc_layer with run_if) redoes the whole layer from the untouched inputs.
// =================================================================================================
// FINAL (the last layer of the last block): the layer's output has one reader, final_conv (Conv1d 32 -> 1, k7).  Instead of writing
// 32 channels for that kernel to read back, the workgroup applies the conv to its own 256 columns: every lane folds its 8 channels
// into 7 per-tap partial sums per column, the four lane groups that share a column meet in LDS (the x image is dead by then), one
// thread per column adds them in a fixed order and stores the sum to eps_acc; the 3 + 3 columns at each tile edge get the missing
// taps from the neighbour tile, both sides with one atomic add onto a zeroed word (two addends: the order cannot change the bits).
// k_final_acc turns eps_acc (+ bias) into eps / the sampler update and leaves it zeroed.
// (wref, the fp32 conv weights, is no longer read here -- the halo columns use the A operand registers -- and stays in the signature
// for the fp32 twin launched with the same argument list.)
// UP = r > 0 (the first layer of a block, DIL = 1): the block's ConvTranspose1d (k_convt_h2<r>) runs inside the staging -- xin is then
// the block's INPUT [B][32][L / r], the up-sampled x never goes to HBM and back (hop 256: 226 MB each way and a 73 us launch).
// Wave w takes output phases w * r/4 ..: the same MFMA sequence on the same operands as k_convt_h2 (bias in the accumulator, per k
// group h.h, h.l, l.h), so x -- and with it everything behind -- keeps its bits.  x + skip then meets in the parking area: skip is
// loaded the coalesced way (wave = channel group, lane = 4 columns) and parked, each lane of the conv's result layout (16 channels of
// one column) reads its 16 skip values back from there, writes x' over them and the leaky-relu pieces into the x image.
// up_flag: the ConvTranspose's own range flag (raised together with the layer's: the host then redoes both on fp32 kernels).
// VAR (option "lvc_variant"; the fused instantiations only).  UP > 0: 1 = the parking area holds the ConvTranspose's x phase-major -- [channel][phase][position], for
// r = 8 with the position XOR-ed by 16 in phases 4..7 -- so that the 32 lanes of a matrix tile (32 positions of ONE phase: columns r
// apart) store to 32 different banks instead of 4-/8-way into 8 / 4 of them, and a lane reads its four columns back as four
// conflict-free dwords; the frame's record is requested in front of the ConvTranspose (as the plain layer does: first of all) and
// the conv weights, which come from L2, behind it.  0 = round 3's form (column-major parking, record requested behind the up-sampler).
// FINAL: 1 = the final_conv weights (4 parts x 8 channels x 8 taps = 1 KB) are copied to LDS by the first wave on its way in and
// read from there in the epilogue; 0 = round 3's form: 16 float4 per lane from L2 behind the LVC's last matrix instruction, where
// every wave of the workgroup waited for them at the end of its life (more registers for them earlier would spill).
template <int HOP, int DIL, bool FINAL, int UP = 0, int VAR = 1>
__global__ void __launch_bounds__(256, 2) k_lvc_h2(const float *__restrict__ xin, const float *__restrict__ skip, float *__restrict__ xout,
                                                   const float *__restrict__ kpack, int layer, const float4 *__restrict__ wpack16,
                                                   const float *__restrict__ wref, const float *__restrict__ cbias,
                                                   int *__restrict__ range_flag, int T, const int *__restrict__ lens,
                                                   float *__restrict__ eps_acc, const float4 *__restrict__ ffuse,
                                                   const float4 *__restrict__ up_pack16, const float *__restrict__ up_bias,
                                                   int *__restrict__ up_flag)
{
    static_assert(!FINAL || HOP == 256, "the fused final conv relies on whole-tile utterance lengths");
    static_assert(UP == 0 || (DIL == 1 && (UP == 4 || UP == 8)), "the fused up-sampler belongs to the first layer of a block");
    constexpr int W = 256, WC = 64, H = (DIL + 1 + 3) & ~3, XC = W + 2 * H, YC = W + 2;
    // fused up-sampler: tile columns -H .. W+H-1 = positions q0 - 1 .. of the block input, UPPAD columns in front of the tile make the
    // first one whole; UPN positions, UPROWS image rows (one more position on either side for the second tap)
    constexpr int UPPAD = UP ? UP : 1, UPN = UP ? (W + 2 * UPPAD) / UPPAD : 0, UPROWS = UPN + 2;
    __shared__ __attribute__((aligned(16))) char xp_img[UP ? UPROWS * 128 : 16];      // leaky_relu(block input) pieces, row = position - (q0 - 2)
    __shared__ float hsk[UP ? fd::C * 2 * H : 1];                                       // the up-sampled x at the 2H halo columns
    constexpr int LT = (HOP == 256) ? 1 : 2;           // row tiles per wave   (hop 256: wave = (row tile, column half))
    constexpr int LN = (HOP == 256) ? 4 : 2;           // column tiles per wave
    // the predicted kernel (HBM, the longest latency) is requested as early as the registers allow: hop 256 (one row tile per
    // wave) before the staging; hop 64 (two row tiles) the first after the staging, the second after the conv
    static_assert(2 * H <= 64, "one halo column per lane");
    __shared__ __attribute__((aligned(16))) char xs[XC * 128];       // lrelu(x + skip) pieces, row = column + H
    __shared__ __attribute__((aligned(16))) char ys[YC * 128];       // first the raw x + skip of the centre (fp32 [32][256]), then
    static_assert(YC * 128 >= fd::C * W * 4, "parking area");        // the conv output pieces, row = column + 1
    const int Ln = T * HOP;
    // (an XCD-contiguous tile order was tried for L2 reuse of the halo columns: no measurable gain, and with ragged batches
    // it leaves the XCDs that own the tail of every utterance idle)
    const int ntile = (T * HOP + W - 1) / W, tile = blockIdx.x;
    const int b = blockIdx.y, w0 = tile * W;
    const int Lnb = frames_of(lens, b, T) * HOP;      // this utterance's own length (ragged batch): every bound below; Ln = row stride
    if (tile >= ntile || w0 >= Lnb || skip_after_previous_overflow(range_flag)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int cw = wave * WC;
    const bool wave_valid = (w0 + cw) < Lnb;
    const int mt0 = (HOP == 256) ? (wave & 1) : 0;
    const int lcw = (HOP == 256) ? 128 * (wave >> 1) : cw;     // first LVC column of this wave
    float mx = 0.0f;                                            // largest operand magnitude seen by this thread
    FD_STAMP(0);

    float4 ka[LT][12];
    float4 bz[LT][4];
    auto load_kernel = [&](int m) {
        if (HOP == 256 || wave_valid) {      // hop 256: utterance lengths are whole tiles, every wave of a live workgroup is valid
            const int f = (w0 + lcw) / HOP;
            const float *rec = kpack + ((int64_t)b * T + f) * fd::KREC;
            const float4 *kp4 = reinterpret_cast<const float4 *>(rec + layer * fd::KLAYER) + 2 * lane;
            const float4 *kb4 = reinterpret_cast<const float4 *>(rec + fd::KW + layer * 64);
#pragma unroll
            for (int i = 0; i < 12; ++i) ka[m][i] = lvc_ld<4>(kp4 + ((mt0 + m) * 6 + (i >> 1)) * 128 + (i & 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) bz[m][j] = kb4[(mt0 + m) * 8 + 2 * j + hi];
        }
    };
    constexpr bool UPNEW = UP > 0 && VAR == 1;
    __shared__ float4 ffs[(FINAL && VAR == 1) ? 64 : 1];
    if constexpr (FINAL && VAR == 1) {
        if (tid < 64) ffs[tid] = ffuse[tid];      // visible to everyone behind the staging barrier
    }
#ifndef FD_LVC_LATE_KERNEL
    if constexpr (HOP == 256 && (UP == 0 || UPNEW)) load_kernel(0);
#endif
    // conv weights: A operand pieces [piece][kg][lane] x 8 fp16, k = 16*kg + 8*hi + e = tap*32 + in
    float4 wa[2][6];
    float4 cb[4];
    float hbias;                             // for the halo outputs
    auto load_conv_weights = [&]() {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int kg = 0; kg < 6; ++kg) wa[p][kg] = wpack16[(p * 6 + kg) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) cb[j] = reinterpret_cast<const float4 *>(cbias)[2 * j + hi];
        hbias = cbias[l31];
    };
    if constexpr (!UPNEW) load_conv_weights();
#ifdef FD_LVC_PAD_LOADS     // probe (tools/ubench): is the layer bound by the CU's memory pipe?  N extra 16 B loads per lane (L2 hits)
    {
        float4 pad_[FD_LVC_PAD_LOADS];
#pragma unroll
        for (int i_ = 0; i_ < FD_LVC_PAD_LOADS; ++i_) pad_[i_] = wpack16[(i_ % 12) * 64 + lane];
#pragma unroll
        for (int i_ = 0; i_ < FD_LVC_PAD_LOADS; ++i_) mx = fmaxf(mx, fminf(pad_[i_].x, 0.0f) * 1e-30f);
    }
#endif

    // ---- stage x + skip.  Centre: wave = channel group of 8, lane = 4 columns, so that one column of a thread is one 16 B
    //      slot per piece.  Halo (2H columns): wave = channel group, lane = one column.  Every wave does the same work.
    //      UP: x is not read but computed -- the block's ConvTranspose lands in the parking area first (below). -----------------
    {
        const float *xr = xin + ((int64_t)b * fd::C + wave * 8) * Ln, *sr = skip + ((int64_t)b * fd::C + wave * 8) * Ln;
        const int g = w0 + 4 * lane;
        const bool ok = g < Lnb;                                     // Lnb is a multiple of 64: a quad is all in or all out
        const int hc = lane, hg = (hc < H) ? w0 - H + hc : w0 + W + hc - H;
        const bool hok = hc < 2 * H && hg >= 0 && hg < Lnb;
        float4 xa[8], sa[8];
        float hx[8], hs[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (UP == 0) xa[c] = ok ? lvc_ld<1>(reinterpret_cast<const float4 *>(xr + (int64_t)c * Ln + g)) : make_float4(0.f, 0.f, 0.f, 0.f);
            sa[c] = ok ? lvc_ld<8>(reinterpret_cast<const float4 *>(sr + (int64_t)c * Ln + g)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (UP == 0) hx[c] = hok ? xr[(int64_t)c * Ln + hg] : 0.0f;
            hs[c] = hok ? sr[(int64_t)c * Ln + hg] : 0.0f;
        }
        if constexpr (UP > 0) {
            // ---- the block's ConvTranspose (see the head of the kernel); skip is on its way from HBM meanwhile -----------------
            constexpr int R = UP, NT = (UPN + 31) / 32, PHW = R / 4;
            const int Lq = Ln / R, Lqb = Lnb / R, q0 = w0 / R;
            {   // (a) the block input as leaky-relu pieces: thread = (8-channel group, image row)
                constexpr int NU = 4 * UPROWS, NK = (NU + 255) / 256;
                float v[NK][8];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int u = k * 256 + tid, cg = u / UPROWS, jj = u - cg * UPROWS, j = q0 - 2 + jj;
                    const bool okp = u < NU && j >= 0 && j < Lqb;
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[k][c] = okp ? xin[((int64_t)b * fd::C + cg * 8 + c) * Lq + j] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int u = k * 256 + tid, cg = u / UPROWS, jj = u - cg * UPROWS;
                    if (u < NU) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) { mx = fmaxf(mx, fabsf(v[k][c])); v[k][c] = lrelu(v[k][c], 0.2f); }
                        float4 ph, pl;
                        split8(v[k], ph, pl);
                        *reinterpret_cast<float4 *>(xp_img + h2_off(jj, cg)) = ph;
                        *reinterpret_cast<float4 *>(xp_img + h2_off(jj, 4 + cg)) = pl;
                    }
                }
            }
            float4 ub[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ub[j] = reinterpret_cast<const float4 *>(up_bias)[2 * j + hi];
            // the first phase's weights are requested in front of the barrier (L2 latency under the wait), the next phase's under the MFMAs
            float4 wun[2][4];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) wun[p][kg] = up_pack16[(((wave * PHW) * 2 + p) * 4 + kg) * 64 + lane];
            __syncthreads();
            // (b) wave = PHW output phases; per phase and 32-position tile the ConvTranspose's 12 MFMAs; x goes to the parking area
            //     ([32][256] fp32 in the y area; the 2H halo columns to hsk), zero outside the utterance like the loads of the other path
            float *park = reinterpret_cast<float *>(ys);
#pragma unroll
            for (int pw = 0; pw < PHW; ++pw) {
                const int ph = wave * PHW + pw;
                float4 wu[2][4];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int kg = 0; kg < 4; ++kg) wu[p][kg] = wun[p][kg];
                if (pw + 1 < PHW) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int kg = 0; kg < 4; ++kg) wun[p][kg] = up_pack16[(((ph + 1) * 2 + p) * 4 + kg) * 64 + lane];
                }
                const int offA = (ph < R / 2) ? 0 : 1, offB = offA - 1;      // sel 0 reads position q + offA, sel 1 position q + offB
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int ql = 32 * t + l31, qc = min(ql, UPN - 1);        // position index in the tile (row qc + 1 of the image)
                    f32x16 ah, al;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { ah[r] = f4c(ub[r >> 2], r & 3); al[r] = 0.0f; }
#pragma unroll
                    for (int kg = 0; kg < 4; ++kg) {          // k = 16*kg + 8*hi + e = sel*32 + i
                        const int row = qc + 1 + ((kg >> 1) ? offB : offA), c2 = kg & 1;
                        const float4 b1 = *reinterpret_cast<const float4 *>(xp_img + h2_off(row, c2 * 2 + hi));
                        const float4 b2 = *reinterpret_cast<const float4 *>(xp_img + h2_off(row, 4 + c2 * 2 + hi));
                        ah = mfma_f16(wu[0][kg], b1, ah);
                        al = mfma_f16(wu[0][kg], b2, al);
                        al = mfma_f16(wu[1][kg], b1, al);
                    }
                    const int col = R * ql + ph - UPPAD;                        // this lane's column of the tile
                    if (ql < UPN && col >= -H && col < W + H) {
                        const int gc = w0 + col;
                        const bool inb = gc >= 0 && gc < Lnb;
                        // centre columns: UPV = 1 phase-major (col = R (ql - 1) + ph: position ql - 1 of phase ph), UPV = 0 as they lie
                        const int pcol = UPNEW ? ph * (W / R) + ((ql - 1) ^ ((R == 8) ? 16 * (ph >> 2) : 0)) : col;
                        float *dst = (col >= 0 && col < W) ? park + pcol : hsk + (col < 0 ? col + H : col - W + H);
                        const int cs = (col >= 0 && col < W) ? W : 2 * H;          // channel stride of the destination
#pragma unroll
                        for (int r = 0; r < 16; ++r)                              // D rows 8j + 4hi + i = register 4j + i
                            dst[(8 * (r >> 2) + 4 * hi + (r & 3)) * cs] = inb ? fmaf(al[r], GX_INV_SCALE, ah[r]) : 0.0f;
                    }
                }
            }
            __syncthreads();
            if constexpr (HOP == 256 && !UPNEW) load_kernel(0);
            if constexpr (UPNEW) {
                load_conv_weights();
                // columns 4 lane + j: r = 4 -> position lane of phase j; r = 8 -> position lane / 2 of phase 4 (lane & 1) + j
                const float *pk = reinterpret_cast<const float *>(ys) + wave * 8 * W +
                                  (R == 4 ? lane : (4 * (lane & 1)) * (W / R) + ((lane >> 1) ^ (16 * (lane & 1))));
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    xa[c] = make_float4(pk[c * W], pk[c * W + (W / R)], pk[c * W + 2 * (W / R)], pk[c * W + 3 * (W / R)]);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) xa[c] = *reinterpret_cast<const float4 *>(ys + ((wave * 8 + c) * W + 4 * lane) * 4);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) hx[c] = hc < 2 * H ? hsk[(wave * 8 + c) * (2 * H) + hc] : 0.0f;
        }
#ifdef FD_LVC_LATE_KERNEL
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HOP == 256 && UP == 0) load_kernel(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            xa[c] = make_float4(xa[c].x + sa[c].x, xa[c].y + sa[c].y, xa[c].z + sa[c].z, xa[c].w + sa[c].w);
            mx = amax4(mx, xa[c]);
            *reinterpret_cast<float4 *>(ys + ((wave * 8 + c) * W + 4 * lane) * 4) = xa[c];      // the residual, parked
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = lrelu(f4c(xa[c], j), 0.2f);
            float4 ph, pl;
            split8(v, ph, pl);
            const int row = H + 4 * lane + j;
            *reinterpret_cast<float4 *>(xs + h2_off(row, wave)) = ph;
            *reinterpret_cast<float4 *>(xs + h2_off(row, 4 + wave)) = pl;
        }
        if (hc < 2 * H) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float t = hx[c] + hs[c];
                mx = fmaxf(mx, fabsf(t));
                v[c] = lrelu(t, 0.2f);
            }
            float4 ph, pl;
            split8(v, ph, pl);
            const int row = (hc < H) ? hc : W + hc;
            *reinterpret_cast<float4 *>(xs + h2_off(row, wave)) = ph;
            *reinterpret_cast<float4 *>(xs + h2_off(row, 4 + wave)) = pl;
        }
    }
    if constexpr (HOP != 256) load_kernel(0);
    __syncthreads();
    FD_STAMP(1);
    // residual values of this lane's outputs: registers, so that ys can take the conv output
    float resid[LN][8 * LT];
    {
#pragma unroll
        for (int nt = 0; nt < LN; ++nt)
#pragma unroll
            for (int m = 0; m < LT; ++m)
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    resid[nt][m * 8 + r] =
                        reinterpret_cast<const float *>(ys)[(16 * (mt0 + m) + (r & 3) + 8 * (r >> 2) + 4 * hi) * W + lcw + nt * 32 + l31];
        __syncthreads();
    }
    FD_STAMP(2);

    // ---- dilated conv on the fp16 pipe; y = lrelu(conv) is split again and written as the B image of the LVC -------------
    if (wave_valid) {
        int xo_[3][2][2];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int row = H + cw + l31 + (tap - 1) * DIL;         // + 32*ct rows: the swizzle term is the same
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) xo_[tap][p][c2] = h2_off(row, p * 4 + c2 * 2 + hi);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            f32x16 ah, al;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ah[r] = f4c(cb[r >> 2], r & 3); al[r] = 0.0f; }
#pragma unroll
            for (int kg = 0; kg < 6; ++kg) {
                const float4 b1 = *reinterpret_cast<const float4 *>(xs + xo_[kg >> 1][0][kg & 1] + ct * 32 * 128);
                const float4 b2 = *reinterpret_cast<const float4 *>(xs + xo_[kg >> 1][1][kg & 1] + ct * 32 * 128);
                ah = mfma_f16(wa[0][kg], b1, ah);
                al = mfma_f16(wa[0][kg], b2, al);
                al = mfma_f16(wa[1][kg], b1, al);
            }
            const int cp = cw + ct * 32 + l31, yrow = cp + 1;
            const bool inside = (w0 + cp) < Lnb;                  // y is zero-padded for the LVC taps (modules.py:240)
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
                *reinterpret_cast<uint2 *>(ys + h2_off(yrow, j) + 8 * hi) = ph;
                *reinterpret_cast<uint2 *>(ys + h2_off(yrow, 4 + j) + 8 * hi) = pl;
            }
        }
    } else {
        // a wave past the end of the signal still owns y columns its left neighbour's taps read: they are zero padding
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) *reinterpret_cast<float4 *>(ys + (cw + 1 + lane) * 128 + s8 * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (HOP != 256) load_kernel(1);
    FD_STAMP(3);
    // ---- the two halo columns (-1 and W) the LVC taps reach, on VALU: waves 0 and 1, one column each.  The weights are the conv's
    //      A operand registers: lane (row l31, half hi) holds, per k group, channels 16 (kg & 1) + 8 hi .. + 7 of tap kg / 2 of
    //      output l31.  Three forms were timed in one session (hop 256 / hop 64, us, B=8, T=864): 24 fp32 FMAs per thread on
    //      weights loaded for the purpose 254.1 / 78.3, this one 250.4 / 74.1, two more conv tiles on the matrix pipe 250.9 / 76.9
    //      (profiles/r03_halo_ab.txt) ---------------------------------------------------------------------------------------
    if (wave < 2) {
        const int c = wave ? W : -1, g = w0 + c;
        const bool ok = g >= 0 && g < Lnb;
        float a1 = 0.0f, a2 = 0.0f;
        if (ok) {
#pragma unroll
            for (int kg = 0; kg < 6; ++kg) halo_dot(xs, H + c + ((kg >> 1) - 1) * DIL, (kg & 1) * 2 + hi, wa[0][kg], wa[1][kg], a1, a2);
        }
        float accv = fmaf(a2, GX_INV_SCALE, a1);
        accv += __shfl_xor(accv, 32, 64);
        if (hi == 0) {
            const float v = ok ? lrelu(accv + hbias, 0.2f) : 0.0f;
            mx = fmaxf(mx, fabsf(v));
            const _Float16 v1 = (_Float16)v, v2 = (_Float16)((v - (float)v1) * GX_SCALE);
            const int yrow = c + 1;
            *reinterpret_cast<_Float16 *>(ys + h2_off(yrow, l31 >> 3) + (l31 & 7) * 2) = v1;
            *reinterpret_cast<_Float16 *>(ys + h2_off(yrow, 4 + (l31 >> 3)) + (l31 & 7) * 2) = v2;
        }
    }
    FD_STAMP(4);
    __syncthreads();
    FD_STAMP(5);
    if (wave_valid) {
        // ---- LVC: A = the frame's predicted kernel (rows gate-paired: register r <-> sigmoid input, r+8 <-> tanh input of
        //      channel 16*mt + drow(r), r < 8), split into pieces here ----------------------------------------------------
        float4 kh[LT][6], kl[LT][6];
#pragma unroll
        for (int m = 0; m < LT; ++m)
#pragma unroll
            for (int kg = 0; kg < 6; ++kg) {
                const float4 &a0 = ka[m][2 * kg], &a1 = ka[m][2 * kg + 1];
                mx = amax4(amax4(mx, a0), a1);
                const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                split8(v, kh[m][kg], kl[m][kg]);
            }
        int yo_[3][2][2];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int row = lcw + l31 + tap;                        // y row = column + 1 + (tap - 1); + 32*nt rows
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) yo_[tap][p][c2] = h2_off(row, p * 4 + c2 * 2 + hi);
        }
        const int64_t ooff = (int64_t)b * fd::C * Ln + (int64_t)(4 * hi) * Ln + w0 + lcw + l31;    // + channel*Ln + nt*32
        float *xo = xout + ooff;
        FD_STAMP(6);
#ifdef FD_LVC_PAD_VALU     // probe (tools/ubench): is the layer bound by instruction issue?  N extra independent VALU instructions here
        {
            float pad_ = mx;
#pragma unroll
            for (int i_ = 0; i_ < FD_LVC_PAD_VALU; ++i_) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(pad_));
            mx = fminf(mx, pad_ * 0.0f + mx);
        }
#endif
        const unsigned Lnu = (unsigned)Ln;
#pragma unroll
        for (int nt = 0; nt < LN; ++nt) {
#pragma unroll
            for (int m = 0; m < LT; ++m) {
                f32x16 ah, al;
#pragma unroll
                for (int r = 0; r < 16; ++r) { ah[r] = f4c(bz[m][r >> 2], r & 3); al[r] = 0.0f; }
#pragma unroll
                for (int kg = 0; kg < 6; ++kg) {
                    const float4 b1 = *reinterpret_cast<const float4 *>(ys + yo_[kg >> 1][0][kg & 1] + nt * 32 * 128);
                    const float4 b2 = *reinterpret_cast<const float4 *>(ys + yo_[kg >> 1][1][kg & 1] + nt * 32 * 128);
                    ah = mfma_f16(kh[m][kg], b1, ah);
                    al = mfma_f16(kh[m][kg], b2, al);
                    al = mfma_f16(kl[m][kg], b1, al);
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int chl = 16 * (mt0 + m) + (r & 3) + 8 * (r >> 2);     // channel minus 4*hi
                    const float zs = fmaf(al[r], GX_INV_SCALE, ah[r]), zt = fmaf(al[r + 8], GX_INV_SCALE, ah[r + 8]);
                    if constexpr (FINAL) resid[nt][m * 8 + r] += gate(zs, zt);
                    else lvc_st<2>(xo + ((unsigned)chl * Lnu + (unsigned)(nt * 32)), resid[nt][m * 8 + r] + gate(zs, zt));
                }
            }
        }
    }
    if constexpr (FINAL) {
        // hop 256: utterance lengths are whole tiles, so every wave of a live workgroup is valid and reaches the barrier
        float *pb = reinterpret_cast<float *>(xs);                   // [part = 2 mt + hi][7 taps][256 columns]
        {
            const int part = 2 * mt0 + hi;
            float fw[8][8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float4 lo4 = (VAR == 1) ? ffs[(part * 8 + r) * 2] : ffuse[(part * 8 + r) * 2], hi4 = (VAR == 1) ? ffs[(part * 8 + r) * 2 + 1] : ffuse[(part * 8 + r) * 2 + 1];
                fw[r][0] = lo4.x; fw[r][1] = lo4.y; fw[r][2] = lo4.z; fw[r][3] = lo4.w; fw[r][4] = hi4.x; fw[r][5] = hi4.y; fw[r][6] = hi4.z;
            }
#pragma unroll
            for (int nt = 0; nt < LN; ++nt)
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    float pk = 0.0f;
#pragma unroll
                    for (int r = 0; r < 8; ++r) pk = fmaf(fw[r][k], resid[nt][r], pk);
                    pb[(part * 7 + k) * W + lcw + nt * 32 + l31] = pk;
                }
        }
        __syncthreads();
        auto column_sum = [&](int t) {      // eps[t] = sum_k w[k] . out[t + k - 3], restricted to this tile's columns
            float e = 0.0f;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int col = t + k - 3;
                if (col >= 0 && col < W) {
#pragma unroll
                    for (int part = 0; part < 4; ++part) e += pb[(part * 7 + k) * W + col];
                }
            }
            return e;
        };
        float *ea = eps_acc + (int64_t)b * Ln + w0;
        const float e = column_sum(tid);
        if (tid >= 3 && tid < W - 3) ea[tid] = e;
        else atomicAdd(ea + tid, e);
        if (tid >= 64 && tid < 70) {         // the taps of the neighbour tiles' edge columns that fall on this tile
            const int j = tid - 64, t = j < 3 ? j - 3 : W + j - 3;
            if (w0 + t >= 0 && w0 + t < Lnb) atomicAdd(ea + t, column_sum(t));
        }
    }
    if (!(mx < GX_LIMIT)) {      // also inf; a NaN operand gives a NaN result on either path
        atomicOr(range_flag, 1);
        if constexpr (UP > 0) atomicOr(up_flag, 1);
    }
    FD_STAMP(7);
}

// eps_acc (the final_conv sums of k_lvc_h2<..., FINAL>) -> eps = sum + bias -> eps_out or the reverse-step update; eps_acc is left
// zeroed for the next step.  If that LVC launch flagged its operands the sums are meaningless: they are only cleared here, and the
// plain k_final behind this launch (run_if) redoes the conv from the fp32 kernel's output.
__global__ void __launch_bounds__(256) k_final_acc(float *__restrict__ eps_acc, const float *__restrict__ bias, float *__restrict__ eps_out,
                                                   float *__restrict__ xstate, const StepParams *params, int sampler, int L,
                                                   int64_t n4_total, const int *__restrict__ lens, const int *__restrict__ overflow)
{
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int Lb = lens ? lens[b] * fd::HOPT : L;
    if (t0 >= Lb) return;
    const int64_t i4 = ((int64_t)b * L + t0) >> 2;
    float4 acc = reinterpret_cast<const float4 *>(eps_acc)[i4];
    reinterpret_cast<float4 *>(eps_acc)[i4] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (*overflow) return;
    const float bv = bias[0];
    acc = make_float4(acc.x + bv, acc.y + bv, acc.z + bv, acc.w + bv);
    if (!sampler) {
        reinterpret_cast<float4 *>(eps_out)[i4] = acc;
    } else {
        const float4 xv = reinterpret_cast<const float4 *>(xstate)[i4];
        reinterpret_cast<float4 *>(xstate)[i4] = fdk::sampler_update4(xv, acc, params, i4, n4_total);
    }
}

// =================================================================================================
// The hop-8 LVC layer (first block: 8 samples per frame), all VALU and one frame per wave.  At hop 8 a 32-column matrix tile
// would straddle four different predicted kernels, and the work is tiny (6912 columns per utterance) but each frame drags a
// 24.8 KB kernel record out of HBM: what matters is how many of those reads are in flight.  Workgroup = 32 columns = 4 frames,
// wave = frame (1728 workgroups at B=8 instead of 432, no loop over frames).  The frame's record is requested first; the
// dilated conv of the wave's own 10 columns (8 + the two the LVC taps reach) runs on VALU from a shared leaky_relu(x+skip)
// window -- lane = (output channel, column half), weights from LDS -- while the record is on its way; y stays
// wave-private in LDS; LVC: lane = output row, gate by a 16-lane shuffle.
// =================================================================================================
template <int DIL>
__global__ void __launch_bounds__(256, 2) k_lvc_h8(const float *__restrict__ xin, const float *__restrict__ skip, float *__restrict__ xout,
                                                   const float *__restrict__ kpack, int layer, const float *__restrict__ wref,
                                                   const float *__restrict__ cbias, int T, const int *__restrict__ lens,
                                                   const int *__restrict__ run_if)
{
    constexpr int HOP = 8, W = 32, H = (DIL + 1 + 3) & ~3, XLD = W + 2 * H, YLD = 12;
    if (run_if && *run_if == 0) return;      // fallback launch behind k_lvc_h8m: only when that kernel flagged its operands
    __shared__ __attribute__((aligned(16))) float xs[fd::C * XLD];          // leaky_relu(x + skip), column c at index c + H
    __shared__ __attribute__((aligned(16))) float xr[fd::C * W];            // raw x + skip of the centre: the residual
    __shared__ __attribute__((aligned(16))) float ys[4][fd::C * YLD];       // per wave: y of columns 8*wave-1 .. 8*wave+8 (+2 pad)
    __shared__ float wl[fd::C * 3 * fd::C];                                 // conv weights as [in*3 + k][out]: lane = out reads row by row
    const int Ln = T * HOP;
    const int b = blockIdx.y, w0 = blockIdx.x * W;
    const int Tb = frames_of(lens, b, T), Lnb = Tb * HOP;
    if (w0 >= Lnb) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int f = w0 / HOP + wave;                  // this wave's frame
    const bool frame_valid = f < Tb;
    // (a) the frame's predicted kernel: lane = output row (mt = hi, row = l31): its k = 16*kg + e and 16*kg + 8 + e shares
    float4 ke[12], ko[12];
    float bzv = 0.0f;
    if (frame_valid) {
        const float *rec = kpack + ((int64_t)b * T + f) * fd::KREC;
        const float4 *kp4 = reinterpret_cast<const float4 *>(rec + layer * fd::KLAYER) + (hi * 6 * 64 + l31) * 2;
#pragma unroll
        for (int i = 0; i < 12; ++i) { ke[i] = kp4[(i >> 1) * 128 + (i & 1)]; ko[i] = kp4[(i >> 1) * 128 + 64 + (i & 1)]; }
        bzv = rec[fd::KW + layer * 64 + lane];
    }
    // (b) conv weights w[o][i][k] (12 KB, L2) -> LDS transposed
    float wst[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { const int idx = k * 256 + tid; wst[k] = wref[(idx & 31) * (fd::C * 3) + (idx >> 5)]; }
    const float cbv = cbias[l31];
    // (c) x + skip with halo: thread = (channel, 4 columns)
    {
        const float *xp = xin + (int64_t)b * fd::C * Ln, *sp = skip + (int64_t)b * fd::C * Ln;
        constexpr int NF4 = XLD / 4, TOTAL = fd::C * NF4, NK = (TOTAL + 255) / 256;
        float4 xa[NK], sa[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4, g = w0 - H + 4 * c4;
            const bool ok = idx < TOTAL && g >= 0 && g < Lnb;
            xa[k] = ok ? *reinterpret_cast<const float4 *>(xp + (int64_t)ci * Ln + g) : make_float4(0.f, 0.f, 0.f, 0.f);
            sa[k] = ok ? *reinterpret_cast<const float4 *>(sp + (int64_t)ci * Ln + g) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 256 + tid, ci = idx / NF4, c4 = idx - ci * NF4;
            if (idx < TOTAL) {
                const float4 r = make_float4(xa[k].x + sa[k].x, xa[k].y + sa[k].y, xa[k].z + sa[k].z, xa[k].w + sa[k].w);
                *reinterpret_cast<float4 *>(xs + ci * XLD + 4 * c4) = make_float4(lrelu(r.x, 0.2f), lrelu(r.y, 0.2f), lrelu(r.z, 0.2f), lrelu(r.w, 0.2f));
                if (c4 >= H / 4 && c4 < H / 4 + W / 4) *reinterpret_cast<float4 *>(xr + ci * W + 4 * c4 - H) = r;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) wl[k * 256 + tid] = wst[k];
    __syncthreads();
    if (!frame_valid) return;
    // ---- dilated conv of columns 8*wave - 1 + {5*hi .. 5*hi + 4}: lane = (output channel l31, column half hi) ----------------
    {
        float acc[5] = {cbv, cbv, cbv, cbv, cbv};
        const float *xb = xs + H + 8 * wave - 1 + 5 * hi;
#pragma unroll 2
        for (int in = 0; in < fd::C; ++in)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float wv = wl[(in * 3 + k) * fd::C + l31];
                const float *xq = xb + in * XLD + (k - 1) * DIL;
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[c] = fmaf(wv, xq[c], acc[c]);
            }
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int g = w0 + 8 * wave - 1 + 5 * hi + c;             // y is zero outside the signal (modules.py:240)
            ys[wave][l31 * YLD + 5 * hi + c] = (g >= 0 && g < Lnb) ? lrelu(acc[c], 0.2f) : 0.0f;
        }
    }
    // the y window is wave-private: the LDS writes above only have to be visible to this wave's own reads below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- LVC: lane = output row (mt = hi, row = l31), 8 columns; y index = column + 1 + (tap - 1) ----------------------------
    float z[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = bzv;
#pragma unroll
    for (int in = 0; in < fd::C; ++in) {
        const float *yr = ys[wave] + in * YLD;
        const float4 y0 = *reinterpret_cast<const float4 *>(yr);
        const float4 y1 = *reinterpret_cast<const float4 *>(yr + 4);
        const float2 y2 = *reinterpret_cast<const float2 *>(yr + 8);
        const float yv[10] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w, y2.x, y2.y};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int kk = tap * 32 + in, kg = kk >> 4, e = kk & 7;
            const float kv = (kk & 8) ? f4c(ko[2 * kg + (e >> 2)], e & 3) : f4c(ke[2 * kg + (e >> 2)], e & 3);
#pragma unroll
            for (int c = 0; c < 8; ++c) z[c] = fmaf(kv, yv[c + tap], z[c]);
        }
    }
    // gate: rows 0..15 of a tile hold the sigmoid inputs, rows 16..31 the tanh inputs of the same channels
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float zt = __shfl_down(z[c], 16, 64);
        z[c] = gate(z[c], zt);
    }
    if ((lane & 16) == 0) {
        const int ch = 16 * hi + (lane & 15), c0 = 8 * wave;
        float *dst = xout + ((int64_t)b * fd::C + ch) * Ln + w0 + c0;
        const float *rr = xr + ch * W + c0;
        *reinterpret_cast<float4 *>(dst) = make_float4(rr[0] + z[0], rr[1] + z[1], rr[2] + z[2], rr[3] + z[3]);
        *reinterpret_cast<float4 *>(dst + 4) = make_float4(rr[4] + z[4], rr[5] + z[5], rr[6] + z[6], rr[7] + z[7]);
    }
}

// =================================================================================================
// The hop-8 layer on the matrix pipe.  A frame is only 8 columns wide, so the 32x32 tiles of the other layers do not fit (one tile
// would straddle four predicted kernels); v_mfma_f32_16x16x32_f16 does: rows = 16 output channels, cols = 16 columns of which a
// frame uses 8, k = one tap x 32 input channels.  With the 2-piece fp16 operands of the rest of the pipe:
//   conv   32 -> 32 channels over the workgroup's 32 columns = four 16x16 tiles, one per wave: 9 MFMAs (3 taps x 3 piece products);
//   LVC    wave = frame: Z[64 x 8] = K_f[64 x 96] Y[96 x 8] = four 16-row tiles x 3 taps x 3 piece products = 36 MFMAs; the frame
//          record's layout ([mt][kg][row32 + 32 g][8], fd_internal.h) already is the A operand of this instruction: lane (r, g4) of
//          tile (mt, half) finds its 8 consecutive k at ((mt*6 + 2 tap + g4/2)*64 + 16 half + r + 32 (g4 & 1))*8; sigmoid and tanh
//          inputs of a channel are the SAME register of the two tiles (mt, 0) and (mt, 1), so the gate stays lane-local.
// ~45 MFMAs of 16 cycles and ~550 VALU instructions per wave instead of ~1600 VALU (k_lvc_h8, which stays as the fp32 fallback).
// =================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(const float4 &a, const float4 &b, f32x4 c)
{
    union { float4 f; f16x8 h; } ua, ub;
    ua.f = a;
    ub.f = b;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.h, ub.h, c, 0, 0, 0);
}

template <int DIL>
__global__ void __launch_bounds__(256, 2) k_lvc_h8m(const float *__restrict__ xin, const float *__restrict__ skip, float *__restrict__ xout,
                                                    const float *__restrict__ kpack, int layer, const float4 *__restrict__ wpack16,
                                                    const float *__restrict__ wref, const float *__restrict__ cbias,
                                                    int *__restrict__ range_flag, int T, const int *__restrict__ lens)
{
    constexpr int HOP = 8, W = 32, H = (DIL + 1 + 3) & ~3, XC = W + 2 * H, NQ = XC / 4;
    static_assert(8 * NQ <= 256, "one (channel quad, column quad) unit per thread");
    __shared__ __attribute__((aligned(16))) char xs[XC * 128];          // leaky_relu(x + skip) pieces, row = column + H
    __shared__ __attribute__((aligned(16))) char ys[(W + 2) * 128];     // conv output pieces, row = column + 1
    __shared__ __attribute__((aligned(16))) float xr[fd::C * W];        // raw x + skip of the centre: the residual
    const int Ln = T * HOP;
    const int b = blockIdx.y, w0 = blockIdx.x * W;
    const int Tb = frames_of(lens, b, T), Lnb = Tb * HOP;
    if (w0 >= Lnb || skip_after_previous_overflow(range_flag)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, g4 = lane >> 4;
    const int f = w0 / HOP + wave;                  // this wave's frame
    const bool frame_valid = f < Tb;
    float mx = 0.0f;
    // (a) the frame's predicted kernel, fp32, in the A-operand order of the 16x16x32 tiles: [mt][half][tap] x 8 consecutive k
    float4 ka[2][2][3][2];
    float4 bz[2][2];
    if (frame_valid) {
        const float *rec = kpack + ((int64_t)b * T + f) * fd::KREC;
        const float4 *kp4 = reinterpret_cast<const float4 *>(rec + layer * fd::KLAYER);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) {
                    const int e8 = (mt * 6 + 2 * tap + (g4 >> 1)) * 64 + 16 * hf + c16 + 32 * (g4 & 1);
                    ka[mt][hf][tap][0] = kp4[2 * e8];
                    ka[mt][hf][tap][1] = kp4[2 * e8 + 1];
                }
                bz[mt][hf] = *reinterpret_cast<const float4 *>(rec + fd::KW + layer * 64 + mt * 32 + 16 * hf + 4 * g4);
            }
    }
    // (b) x + skip with halo: thread = (channel quad, column quad)
    {
        const int q = tid / NQ, c4 = tid - q * NQ, g = w0 - H + 4 * c4;
        const bool unit = tid < 8 * NQ, ok = unit && g >= 0 && g < Lnb;      // (Lnb, w0, H are multiples of 4: a quad is all in or all out)
        const float *xp = xin + ((int64_t)b * fd::C + 4 * q) * Ln, *sp = skip + ((int64_t)b * fd::C + 4 * q) * Ln;
        float4 xa[4], sa[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            xa[c] = ok ? *reinterpret_cast<const float4 *>(xp + (int64_t)c * Ln + g) : make_float4(0.f, 0.f, 0.f, 0.f);
            sa[c] = ok ? *reinterpret_cast<const float4 *>(sp + (int64_t)c * Ln + g) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (unit) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                xa[c] = make_float4(xa[c].x + sa[c].x, xa[c].y + sa[c].y, xa[c].z + sa[c].z, xa[c].w + sa[c].w);
                if (c4 >= H / 4 && c4 < H / 4 + W / 4) *reinterpret_cast<float4 *>(xr + (4 * q + c) * W + 4 * c4 - H) = xa[c];
            }
            const int slot = q >> 1, half8 = (q & 1) * 8;       // channels 4q .. 4q+3 = half of the 16 B slot q/2
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { v[c] = lrelu(f4c(xa[c], j), 0.2f); mx = fmaxf(mx, fabsf(v[c])); }
                uint2 ph, pl;
                split2(v[0], v[1], ph.x, pl.x);
                split2(v[2], v[3], ph.y, pl.y);
                const int row = 4 * c4 + j;
                *reinterpret_cast<uint2 *>(xs + h2_off(row, slot) + half8) = ph;
                *reinterpret_cast<uint2 *>(xs + h2_off(row, 4 + slot) + half8) = pl;
            }
        }
    }
    // (c) conv weights of this wave's 16-row tile: A operand pieces [row tile][tap][piece][lane] x 8 fp16 (L2)
    const int rt = wave & 1, ctl = wave >> 1;
    float4 wa[3][2];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int p = 0; p < 2; ++p) wa[tap][p] = wpack16[((rt * 3 + tap) * 2 + p) * 64 + lane];
    const float4 cb4 = *reinterpret_cast<const float4 *>(cbias + 16 * rt + 4 * g4);
    const int hside = tid >> 7, ho = (tid & 127) >> 2, hq = tid & 3;
    float4 hwt[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) hwt[j] = reinterpret_cast<const float4 *>(wref + (ho * fd::C + 8 * hq) * 3)[j];
    const float hbias = cbias[ho];
    __syncthreads();
    // ---- dilated conv: wave = (16 output channels, 16 columns); y = leaky_relu(conv) goes to the y image as pieces ----------------
    {
        f32x4 ah = {cb4.x, cb4.y, cb4.z, cb4.w}, al = {0.f, 0.f, 0.f, 0.f};
        const int col = 16 * ctl + c16;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int row = H + col + (tap - 1) * DIL;
            const float4 b1 = *reinterpret_cast<const float4 *>(xs + h2_off(row, g4));
            const float4 b2 = *reinterpret_cast<const float4 *>(xs + h2_off(row, 4 + g4));
            ah = mfma16(wa[tap][0], b1, ah);
            al = mfma16(wa[tap][0], b2, al);
            al = mfma16(wa[tap][1], b1, al);
        }
        const bool inside = (w0 + col) < Lnb;                  // y is zero outside the signal (modules.py:240)
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = inside ? lrelu(fmaf(al[r], GX_INV_SCALE, ah[r]), 0.2f) : 0.0f;
            mx = fmaxf(mx, fabsf(v[r]));
        }
        uint2 ph, pl;
        split2(v[0], v[1], ph.x, pl.x);
        split2(v[2], v[3], ph.y, pl.y);
        // D rows 4 g4 + r of row tile rt = channels 16 rt + 4 g4 + r: half of slot 2 rt + g4 / 2
        *reinterpret_cast<uint2 *>(ys + h2_off(col + 1, 2 * rt + (g4 >> 1)) + (g4 & 1) * 8) = ph;
        *reinterpret_cast<uint2 *>(ys + h2_off(col + 1, 4 + 2 * rt + (g4 >> 1)) + (g4 & 1) * 8) = pl;
    }
    // ---- the two halo columns (-1 and W) the LVC taps reach: VALU on the reassembled x image, 4 threads per output ----------------
    {
        const int c = hside ? W : -1, g = w0 + c;
        const bool ok = g >= 0 && g < Lnb;
        float accv = 0.0f;
        if (ok) {
            const float wv[24] = {hwt[0].x, hwt[0].y, hwt[0].z, hwt[0].w, hwt[1].x, hwt[1].y, hwt[1].z, hwt[1].w,
                                  hwt[2].x, hwt[2].y, hwt[2].z, hwt[2].w, hwt[3].x, hwt[3].y, hwt[3].z, hwt[3].w,
                                  hwt[4].x, hwt[4].y, hwt[4].z, hwt[4].w, hwt[5].x, hwt[5].y, hwt[5].z, hwt[5].w};
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int row = H + c + (tap - 1) * DIL;
                union { float4 f; _Float16 h[8]; } p1, p2;
                p1.f = *reinterpret_cast<const float4 *>(xs + h2_off(row, hq));
                p2.f = *reinterpret_cast<const float4 *>(xs + h2_off(row, 4 + hq));
#pragma unroll
                for (int j = 0; j < 8; ++j) accv += wv[j * 3 + tap] * fmaf((float)p2.h[j], GX_INV_SCALE, (float)p1.h[j]);
            }
        }
        accv += __shfl_xor(accv, 1, 64);
        accv += __shfl_xor(accv, 2, 64);
        if (hq == 0) {
            const float v = ok ? lrelu(accv + hbias, 0.2f) : 0.0f;
            mx = fmaxf(mx, fabsf(v));
            const _Float16 v1 = (_Float16)v, v2 = (_Float16)((v - (float)v1) * GX_SCALE);
            const int yrow = c + 1;
            *reinterpret_cast<_Float16 *>(ys + h2_off(yrow, ho >> 3) + (ho & 7) * 2) = v1;
            *reinterpret_cast<_Float16 *>(ys + h2_off(yrow, 4 + (ho >> 3)) + (ho & 7) * 2) = v2;
        }
    }
    __syncthreads();
    if (frame_valid) {
        // ---- LVC of this wave's frame: columns 8 wave .. 8 wave + 7 are MFMA columns 0..7 (columns 8..15 repeat column 7: never stored)
        const int ycol = 8 * wave + min(c16, 7);
        float4 yb[3][2];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            yb[tap][0] = *reinterpret_cast<const float4 *>(ys + h2_off(ycol + tap, g4));          // y row = column + 1 + (tap - 1)
            yb[tap][1] = *reinterpret_cast<const float4 *>(ys + h2_off(ycol + tap, 4 + g4));
        }
        float *xo = xout + ((int64_t)b * fd::C + 4 * g4) * Ln + w0 + 8 * wave + c16;
        const float *rr = xr + (4 * g4) * W + 8 * wave + c16;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 zh[2], zl[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                zh[hf] = f32x4{bz[mt][hf].x, bz[mt][hf].y, bz[mt][hf].z, bz[mt][hf].w};
                zl[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) {
                    const float4 &a0 = ka[mt][hf][tap][0], &a1 = ka[mt][hf][tap][1];
                    mx = amax4(amax4(mx, a0), a1);
                    const float kv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    float4 kh, kl;
                    split8(kv, kh, kl);
                    zh[hf] = mfma16(kh, yb[tap][0], zh[hf]);
                    zl[hf] = mfma16(kh, yb[tap][1], zl[hf]);
                    zl[hf] = mfma16(kl, yb[tap][0], zl[hf]);
                }
            }
            if (c16 < 8) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {      // channel 16 mt + 4 g4 + r: sigmoid input in tile (mt, 0), tanh input in tile (mt, 1)
                    const float zs = fmaf(zl[0][r], GX_INV_SCALE, zh[0][r]), zt = fmaf(zl[1][r], GX_INV_SCALE, zh[1][r]);
                    lvc_st<16>(xo + (int64_t)(16 * mt + r) * Ln, rr[(16 * mt + r) * W] + gate(zs, zt));
                }
            }
        }
    }
    if (!(mx < GX_LIMIT)) atomicOr(range_flag, 1);
}

// =================================================================================================
// a10 + sampler: final_conv Conv1d(32,1,k7) (FastDiff_model.py:67-68,100) with the reverse-step update
// (util.py:219-229) fused into its epilogue.  VALU; each thread produces 4 consecutive samples.
// =================================================================================================
__global__ void __launch_bounds__(256) k_final(const float *__restrict__ x32, const float *__restrict__ w,
                                               const float *__restrict__ bias, float *__restrict__ eps_out,
                                               float *__restrict__ xstate, const StepParams *params, int sampler, int L,
                                               int64_t n4_total, const int *__restrict__ lens, const int *__restrict__ run_if)
{
    if (run_if && *run_if == 0) return;      // fallback launch behind k_final_acc: only when the fused last layer flagged its operands
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int Lb = lens ? lens[b] * fd::HOPT : L;          // this utterance's own length (ragged batch)
    if (t0 >= Lb) return;
    const float bv = bias[0];
    float4 acc = make_float4(bv, bv, bv, bv);
#pragma unroll 8
    for (int ci = 0; ci < fd::C; ++ci) {
        const float *xr = x32 + ((int64_t)b * fd::C + ci) * L;
        float v[12];
        const float4 m = *reinterpret_cast<const float4 *>(xr + t0);
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi4 = lo;
        if (t0 >= 4) lo = *reinterpret_cast<const float4 *>(xr + t0 - 4);
        if (t0 + 4 < Lb) hi4 = *reinterpret_cast<const float4 *>(xr + t0 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = m.x; v[5] = m.y; v[6] = m.z; v[7] = m.w;
        v[8] = hi4.x; v[9] = hi4.y; v[10] = hi4.z; v[11] = hi4.w;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float wv = w[ci * 7 + k];       // tap k reads x[t + k - 3] = v[(t - t0) + k + 1]
            acc.x += wv * v[k + 1]; acc.y += wv * v[k + 2]; acc.z += wv * v[k + 3]; acc.w += wv * v[k + 4];
        }
    }
    const int64_t i4 = ((int64_t)b * L + t0) >> 2;
    if (!sampler) {
        reinterpret_cast<float4 *>(eps_out)[i4] = acc;
    } else {
        const float4 xv = reinterpret_cast<const float4 *>(xstate)[i4];
        reinterpret_cast<float4 *>(xstate)[i4] = fdk::sampler_update4(xv, acc, params, i4, n4_total);
    }
}

}  // namespace fdk_fast

// ------------------------------------------------------------------------------------------------
// stage drivers (fast mode)
// ------------------------------------------------------------------------------------------------
namespace fdk {
using namespace fdk_fast;

hipError_t fast_first_conv(const Launch &L, const StepIO &io, int B, int T)
{
    const DevWeights &w = L.ctx->w;
    const int Lf = T * fd::HOPT;
    fd_context *c = L.ctx;
    const bool adv = io.sampler && c->advance_pending;      // the previous step of this sequence left its bookkeeping to us
#define FD_FIRST(V_) FD_LAUNCH(L, "first_conv", k_first_conv<V_>, dim3((Lf + 1023) / 1024, B), dim3(256), 0, io.x_in, w.first.w, w.first.b, \
                               c->ws.a[0], Lf, c->step_lens, adv ? c->ws.params : (StepParams *)nullptr, c->ws.range_flag)
    switch (c->first_variant) {
    case 4: FD_FIRST(4); break;
    case 5: FD_FIRST(5); break;
    case 2: FD_FIRST(2); break;
    case 3: FD_FIRST(3); break;
    case 0: FD_FIRST(0); break;
    default: FD_FIRST(1); break;
    }
#undef FD_FIRST
    if (adv) c->advance_pending = false;
    return hipSuccess;
}

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
        FD_LAUNCH(L, name, k_kp_front_h2, grid, dim3(256), 0, io.mel, c->ws.kp_hB, reinterpret_cast<char *>(c->ws.h_f16), k2,
                  (const float *)c->ws.noise, (const StepParams *)c->ws.params, io.sampler, B, T, gx_rows(T), c->ws.range_flag, c->step_lens);
        c->h_image_ready = true;      // the GEMM's fp16 image of h is written (k_h_split not needed)
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

hipError_t fast_kp_gemm(const Launch &L, int B, int T, int blk0, int nblk, int wg_per_cu)
{
    // blk0, nblk: the blocks to compute (the fp16x2 kernel only: the fp32 kernel, whole job or early-exit fallback, always covers all
    // three and is launched with the range that starts at block 0); wg_per_cu: persistent workgroups per CU (2 alone on the chip)
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    const int tiles_per_utt = (T + 31) / 32;
    const int chunks_per_utt = (tiles_per_utt + GEMM_CT - 1) / GEMM_CT;
    const int chunk_tiles = (tiles_per_utt + chunks_per_utt - 1) / chunks_per_utt;     // balanced, <= GEMM_CT
    const int n_items = fd::NBLK * (fd::KREC / 128) * B * chunks_per_utt;
    const int grid = n_items < 2 * c->num_cus ? n_items : 2 * c->num_cus;              // persistent: 2 workgroups per CU
    const Pipe pipe = fd_pipe(c, c->gemm_f16 && w.gemm_f16_ok, 0);
    const bool f16 = pipe != PIPE_F32_ONLY;
    if (f16) {
        const int R = gx_rows(T);
        const int chunks = (T + GX_CT * 32 - 1) / (GX_CT * 32), items = nblk * (fd::KREC / 128) * B * chunks;
        const int grid2 = items < wg_per_cu * c->num_cus ? items : wg_per_cu * c->num_cus;
        if (!c->h_image_ready && blk0 == 0)      // the fp16-pipe predictor front writes the image itself
            FD_LAUNCH(L, "h_split", k_h_split, dim3((32 * R + 255) / 256, fd::NBLK * B), dim3(256), 0, (const float *)c->ws.kp_hB,
                      reinterpret_cast<unsigned *>(c->ws.h_f16), c->ws.range_flag, B, T, R, c->step_lens);
        FD_LAUNCH(L, "kp_gemm_f16x2", k_kp_gemm_h2, dim3(grid2), dim3(256), 0, reinterpret_cast<const char *>(c->ws.h_f16), c->ws.kpack,
                  reinterpret_cast<const float4 *>(w.gemm_h2_pack[0]), reinterpret_cast<const float4 *>(w.gemm_h2_pack[1]),
                  reinterpret_cast<const float4 *>(w.gemm_h2_pack[2]), w.gemm_bias[0], w.gemm_bias[1], w.gemm_bias[2],
                  (const int *)c->ws.range_flag, B, T, R, chunks, items, c->step_lens, blk0, nblk);
        if (pipe == PIPE_F16_ONLY || blk0 != 0) return hipSuccess;
    }
    if (blk0 != 0) return hipSuccess;
    // fp32 matrix pipe: the whole job when the fp16 form is off, otherwise an early-exit launch that only works when
    // k_h_split found operands outside the fp16 range
    FD_LAUNCH(L, f16 ? "kp_gemm_fp32_fallback" : "kp_gemm", k_kp_gemm, dim3(grid), dim3(256), 0, (const float *)c->ws.kp_hB, c->ws.kpack,
              w.gemm_pack[0], w.gemm_pack[1], w.gemm_pack[2], w.gemm_bias[0], w.gemm_bias[1], w.gemm_bias[2], B, T, chunks_per_utt,
              chunk_tiles, n_items, f16 ? (const int *)c->ws.range_flag : (const int *)nullptr, c->step_lens);
    return hipSuccess;
}

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

template <int HOP, int DIL>
static hipError_t launch_lvc(const Launch &L, const char *name, int n, int layer, const float *x_in, const float *skip, float *x_out,
                             int B, int T, bool up)
{
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    constexpr int W = LvcCfg<HOP, DIL>::W;
    const int Ln = T * HOP;
    // block n's records; with a hoisted predictor (fd_internal.h) the batch behind kpack is hoist_np * B entries and this step's are
    // the hoist_step-th B of them
    const float *kp = c->ws.kpack + ((int64_t)n * c->hoist_np + c->hoist_step) * B * T * fd::KREC;
    const int *run_if = nullptr;
    if constexpr (HOP == 256 && DIL == 27) c->final_fused = false;
    if constexpr (HOP >= 64 && DIL == 1) {
        if (up) {      // x_in = the block's input: the ConvTranspose runs inside the layer (the caller made sure both stages are fp16x2-only)
            constexpr int R = (HOP == 256) ? 4 : 8;
            if (c->lvc_variant == 1)
                FD_LAUNCH(L, HOP == 256 ? "lvc_up_h256" : "lvc_up_h64", (k_lvc_h2<HOP, 1, false, R, 1>), dim3(((Ln + 255) / 256 + 7) / 8 * 8, B), dim3(256), 0, x_in, skip, x_out, kp, layer,
                          reinterpret_cast<const float4 *>(w.lvc_conv_h2[n][layer]), w.blk[n].convs[layer].w, w.blk[n].convs[layer].b,
                          c->ws.range_flag + 1 + n * fd::LAYERS + layer, T, c->step_lens, (float *)nullptr, (const float4 *)nullptr,
                          reinterpret_cast<const float4 *>(w.up_h2[n]), w.blk[n].up.b, c->ws.range_flag + 16 + n);
            else
                FD_LAUNCH(L, HOP == 256 ? "lvc_up_h256" : "lvc_up_h64", (k_lvc_h2<HOP, 1, false, R, 0>), dim3(((Ln + 255) / 256 + 7) / 8 * 8, B), dim3(256), 0, x_in, skip, x_out, kp, layer,
                          reinterpret_cast<const float4 *>(w.lvc_conv_h2[n][layer]), w.blk[n].convs[layer].w, w.blk[n].convs[layer].b,
                          c->ws.range_flag + 1 + n * fd::LAYERS + layer, T, c->step_lens, (float *)nullptr, (const float4 *)nullptr,
                          reinterpret_cast<const float4 *>(w.up_h2[n]), w.blk[n].up.b, c->ws.range_flag + 16 + n);
            return hipSuccess;
        }
    }
    if constexpr (HOP >= 64) {
        const Pipe pipe = fd_pipe(c, c->lvc_f16 && w.lvc_f16_ok, 1 + n * fd::LAYERS + layer);
        if (pipe != PIPE_F32_ONLY) {
            int *flag = c->ws.range_flag + 1 + n * fd::LAYERS + layer;
            // the last layer of the last block feeds final_conv only: fused unless someone wants to look at the block output
            c->final_fused = false;
            if constexpr (HOP == 256 && DIL == 27) c->final_fused = c->fast[ST_FINAL] && !c->keep_taps && c->fuse_final;
            if constexpr (HOP == 256 && DIL == 27) {
                if (c->final_fused && c->lvc_variant == 1)      // its own profile row: this variant never writes its 32 output channels
                    FD_LAUNCH(L, "lvc_final_h256", (k_lvc_h2<HOP, DIL, true, 0, 1>), dim3(((Ln + 255) / 256 + 7) / 8 * 8, B), dim3(256), 0, x_in, skip, x_out, kp,
                              layer, reinterpret_cast<const float4 *>(w.lvc_conv_h2[n][layer]), w.blk[n].convs[layer].w,
                              w.blk[n].convs[layer].b, flag, T, c->step_lens, c->ws.eps_acc, reinterpret_cast<const float4 *>(w.final_fuse),
                              (const float4 *)nullptr, (const float *)nullptr, (int *)nullptr);
                else if (c->final_fused)
                    FD_LAUNCH(L, "lvc_final_h256", (k_lvc_h2<HOP, DIL, true, 0, 0>), dim3(((Ln + 255) / 256 + 7) / 8 * 8, B), dim3(256), 0, x_in, skip, x_out, kp,
                              layer, reinterpret_cast<const float4 *>(w.lvc_conv_h2[n][layer]), w.blk[n].convs[layer].w,
                              w.blk[n].convs[layer].b, flag, T, c->step_lens, c->ws.eps_acc, reinterpret_cast<const float4 *>(w.final_fuse),
                              (const float4 *)nullptr, (const float *)nullptr, (int *)nullptr);
            }
            if (!c->final_fused)
                FD_LAUNCH(L, name, (k_lvc_h2<HOP, DIL, false>), dim3(((Ln + 255) / 256 + 7) / 8 * 8, B), dim3(256), 0, x_in, skip, x_out, kp,
                          layer, reinterpret_cast<const float4 *>(w.lvc_conv_h2[n][layer]), w.blk[n].convs[layer].w,
                          w.blk[n].convs[layer].b, flag, T, c->step_lens, (float *)nullptr, (const float4 *)nullptr,
                          (const float4 *)nullptr, (const float *)nullptr, (int *)nullptr);
            run_if = flag;
            name = "lvc_fp32_fallback";
            if (pipe == PIPE_F16_ONLY) return hipSuccess;
        }
    }
    FD_LAUNCH(L, name, (k_lvc_layer<HOP, DIL>), dim3((Ln + W - 1) / W, B), dim3(256), 0, x_in, skip, x_out, kp, layer,
              w.lvc_conv_pack[n][layer], w.blk[n].convs[layer].w, w.blk[n].convs[layer].b, T, run_if, c->step_lens);
    return hipSuccess;
}

hipError_t fast_lvc_layer(const Launch &L, int n, int layer, const float *x_in, const float *skip, float *x_out, int B, int T, bool up)
{
#define FD_LVC_CASE(HOP_, NAME_)                                                                                    \
    switch (layer) {                                                                                                \
    case 0: return launch_lvc<HOP_, 1>(L, NAME_ "_d1", n, layer, x_in, skip, x_out, B, T, up);                           \
    case 1: return launch_lvc<HOP_, 3>(L, NAME_ "_d3", n, layer, x_in, skip, x_out, B, T, up);                           \
    case 2: return launch_lvc<HOP_, 9>(L, NAME_ "_d9", n, layer, x_in, skip, x_out, B, T, up);                           \
    default: return launch_lvc<HOP_, 27>(L, NAME_ "_d27", n, layer, x_in, skip, x_out, B, T, up);                        \
    }
    if (n == 0) {
        fd_context *c = L.ctx;
        const DevWeights &w = c->w;
        const int Ln = T * 8;
        const float *kp = c->ws.kpack + (int64_t)c->hoist_step * B * T * fd::KREC;      // (block 0; hoisted predictor: this step's entries)
        const dim3 grid((Ln + 31) / 32, B);
        const Pipe pipe = fd_pipe(c, c->lvc_f16 && w.lvc_f16_ok && c->lvc_h8_mfma, 1 + layer);
        int *flag = c->ws.range_flag + 1 + layer;
#define FD_H8(DIL_, NAME_)                                                                                          \
        if (pipe != PIPE_F32_ONLY)                                                                                   \
            FD_LAUNCH(L, NAME_, k_lvc_h8m<DIL_>, grid, dim3(256), 0, x_in, skip, x_out, kp, layer,                     \
                      reinterpret_cast<const float4 *>(w.lvc_conv_h16[layer]), w.blk[0].convs[layer].w, w.blk[0].convs[layer].b, flag, T, \
                      c->step_lens);                                                                                 \
        if (pipe != PIPE_F16_ONLY)                                                                                   \
            FD_LAUNCH(L, pipe == PIPE_F32_ONLY ? NAME_ : "lvc_fp32_fallback", k_lvc_h8<DIL_>, grid, dim3(256), 0, x_in, skip, x_out, kp, layer, \
                      w.blk[0].convs[layer].w, w.blk[0].convs[layer].b, T, c->step_lens, pipe == PIPE_F32_ONLY ? (const int *)nullptr : (const int *)flag)
        switch (layer) {
        case 0: FD_H8(1, "lvc_layer_h8_d1"); break;
        case 1: FD_H8(3, "lvc_layer_h8_d3"); break;
        case 2: FD_H8(9, "lvc_layer_h8_d9"); break;
        default: FD_H8(27, "lvc_layer_h8_d27"); break;
        }
#undef FD_H8
        return hipSuccess;
    }
    if (n == 1) { FD_LVC_CASE(64, "lvc_layer_h64") }
    FD_LVC_CASE(256, "lvc_layer_h256")
#undef FD_LVC_CASE
}

hipError_t fast_final(const Launch &L, const StepIO &io, const float *x32, int B, int T)
{
    fd_context *c = L.ctx;
    const DevWeights &w = c->w;
    const int Lf = T * fd::HOPT;
    const int *run_if = nullptr;
    const char *name = "final_conv_update";
    if (c->final_fused) {       // the last LVC layer already left the conv sums in eps_acc
        const int *flag = c->ws.range_flag + 1 + 2 * fd::LAYERS + 3;
        FD_LAUNCH(L, "final_update", k_final_acc, dim3((Lf + 1023) / 1024, B), dim3(256), 0, c->ws.eps_acc, w.final_.b, io.eps_out, c->ws.x,
                  (const StepParams *)c->ws.params, io.sampler, Lf, (int64_t)B * Lf / 4, c->step_lens, flag);
        run_if = flag;
        name = "final_conv_fallback";
        c->final_fused = false;
        if (!c->inline_fallback) return hipSuccess;      // fallback = host: a flagged last layer is redone from the host
    }
    FD_LAUNCH(L, name, k_final, dim3((Lf + 1023) / 1024, B), dim3(256), 0, x32, w.final_.w, w.final_.b, io.eps_out,
              c->ws.x, (const StepParams *)c->ws.params, io.sampler, Lf, (int64_t)B * Lf / 4, c->step_lens, run_if);
    return hipSuccess;
}

}  // namespace fdk
