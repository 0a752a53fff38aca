// fd_kernels_lvc.hip -- a7 + a8 + a9 one whole TimeAware LVC layer per launch: skip add, dilated conv, location-variable convolution, gate, residual (modules.py:208-253)
// (one stage of the gfx950 kernel set; shared device helpers: fd_kernels_common.h; the one-thread-per-output twins: fd_kernels_naive.hip)
#include "fd_kernels_common.h"

namespace fdk_fast {

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
#define FD_STAMP_AT(i) do { if (threadIdx.x == 0) { long long *q_ = fd_tl + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 10; \
        q_[i] = (long long)__builtin_amdgcn_s_memrealtime(); \
        if ((i) == 0) { q_[8] = __builtin_amdgcn_s_getreg(63492); q_[9] = __builtin_amdgcn_s_getreg(63508); } } } while (0)
#if defined(FD_TL_VARIANT)
// the fused variants' own phases (round 6): slots 1..4 go to FD_STAMP_X(1..4) inside the up-sampler / the final-conv fold, the regular
// stamps keep 0, 1 (-> slot 5: staging barrier), 6, 7
#define FD_STAMP(i) do { if ((i) == 0 || (i) == 6 || (i) == 7) FD_STAMP_AT(i); else if ((i) == 1) FD_STAMP_AT(5); } while (0)
#define FD_STAMP_X(i) FD_STAMP_AT(i)
#else
#define FD_STAMP(i) FD_STAMP_AT(i)
#endif
#else
#define FD_STAMP(i)
#endif
#ifndef FD_STAMP_X
#define FD_STAMP_X(i)
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
// The parking area holds the ConvTranspose's x phase-major -- [channel][phase][position], for r = 8 with the position XOR-ed by 16 in
// phases 4..7 -- so that the 32 lanes of a matrix tile (32 positions of ONE phase: columns r apart) store to 32 different banks instead
// of 4-/8-way into 8 / 4 of them, and a lane reads its four columns back as four conflict-free dwords; the frame's record is requested
// in front of the ConvTranspose (as the plain layer does: first of all) and the conv weights, which come from L2, behind it.
// FINAL: the final_conv weights (4 parts x 8 channels x 8 taps = 1 KB) are copied to LDS by the first wave on its way in and read from
// there in the epilogue (16 float4 per lane from L2 behind the LVC's last matrix instruction made every wave wait at the end of its life).
template <int HOP, int DIL, bool FINAL, int UP = 0>
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
    const int ntile = (T * HOP + W - 1) / W, tile = lvc_tile_of_workgroup();      // runs of tiles per XCD (fd_kernels_common.h)
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
    __shared__ float4 ffs[FINAL ? 64 : 1];
    if constexpr (FINAL) {
        if (tid < 64) ffs[tid] = ffuse[tid];      // visible to everyone behind the staging barrier
    }
#ifndef FD_LVC_LATE_KERNEL
    if constexpr (HOP == 256) load_kernel(0);
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
    if constexpr (UP == 0) load_conv_weights();
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
#ifdef FD_LVC_PROBE_NOLOADX   // probe (tools/ubench, pair-fusion bound): x and skip already on chip -- values made from the lane number, no loads
            const float pv_ = __int_as_float(0x3c000000 | (lane << 8) | (c << 4));
            if constexpr (UP == 0) xa[c] = make_float4(pv_, -pv_, pv_ * 0.5f, pv_ * 0.25f);
            sa[c] = make_float4(pv_ * 0.125f, pv_, -pv_, pv_ * 2.0f);
#else
            if constexpr (UP == 0) xa[c] = ok ? lvc_ld<1>(reinterpret_cast<const float4 *>(xr + (int64_t)c * Ln + g)) : make_float4(0.f, 0.f, 0.f, 0.f);
            sa[c] = ok ? lvc_ld<8>(reinterpret_cast<const float4 *>(sr + (int64_t)c * Ln + g)) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#ifdef FD_LVC_PROBE_NOLOADX
            if constexpr (UP == 0) hx[c] = __int_as_float(0x3b000000 | (lane << 8) | (c << 4));
            hs[c] = hok ? 0.03125f : 0.0f;
#else
            if constexpr (UP == 0) hx[c] = hok ? xr[(int64_t)c * Ln + hg] : 0.0f;
            hs[c] = hok ? sr[(int64_t)c * Ln + hg] : 0.0f;
#endif
        }
#ifdef FD_LVC_PROBE_UP_NOCONVT   // probe (round 6 gate): the fused up-sampler's own cost -- x made from the lane number, no ConvTranspose phase at all
        if constexpr (UP > 0) {
            load_conv_weights();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float pv_ = __int_as_float(0x3c000000 | (lane << 8) | (c << 4));
                xa[c] = make_float4(pv_, -pv_, pv_ * 0.5f, pv_ * 0.25f);
                hx[c] = pv_ * 0.125f;
            }
            (void)xp_img; (void)hsk; (void)up_pack16; (void)up_bias;
        }
#else
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
            FD_STAMP_X(1);
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
                // (round 6, measured and not kept: the tile's 8 B operands requested one tile ahead + the frame record requested behind
                // this phase to free its 64 registers -- the phase stayed at 3.1 us of a workgroup's life, the layer 243 -> 248 us: the
                // phase is issue time shared with the CU's other workgroup, not LDS latency; LABBOOK R6.5)
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
                        // centre columns phase-major (col = R (ql - 1) + ph: position ql - 1 of phase ph)
                        const int pcol = ph * (W / R) + ((ql - 1) ^ ((R == 8) ? 16 * (ph >> 2) : 0));
                        float *dst = (col >= 0 && col < W) ? park + pcol : hsk + (col < 0 ? col + H : col - W + H);
                        const int cs = (col >= 0 && col < W) ? W : 2 * H;          // channel stride of the destination
#pragma unroll
                        for (int r = 0; r < 16; ++r)                              // D rows 8j + 4hi + i = register 4j + i
                            dst[(8 * (r >> 2) + 4 * hi + (r & 3)) * cs] = inb ? fmaf(al[r], GX_INV_SCALE, ah[r]) : 0.0f;
                    }
                }
            }
            FD_STAMP_X(2);
            __syncthreads();
            FD_STAMP_X(3);
            load_conv_weights();
            {   // columns 4 lane + j: r = 4 -> position lane of phase j; r = 8 -> position lane / 2 of phase 4 (lane & 1) + j
                const float *pk = reinterpret_cast<const float *>(ys) + wave * 8 * W +
                                  (R == 4 ? lane : (4 * (lane & 1)) * (W / R) + ((lane >> 1) ^ (16 * (lane & 1))));
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    xa[c] = make_float4(pk[c * W], pk[c * W + (W / R)], pk[c * W + 2 * (W / R)], pk[c * W + 3 * (W / R)]);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) hx[c] = hc < 2 * H ? hsk[(wave * 8 + c) * (2 * H) + hc] : 0.0f;
            FD_STAMP_X(4);
        }
#endif
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
#ifdef FD_LVC_PROBE_NOSTORE   // probe (pair-fusion bound): the first layer of a pair keeps its output on chip
                    else { const float keep_ = resid[nt][m * 8 + r] + gate(zs, zt); asm volatile("" :: "v"(keep_)); (void)xo; (void)Lnu; (void)chl; }
#else
                    else lvc_st<2>(xo + ((unsigned)chl * Lnu + (unsigned)(nt * 32)), resid[nt][m * 8 + r] + gate(zs, zt));
#endif
                }
            }
        }
    }
#ifdef FD_LVC_PROBE_FINAL_NOFOLD   // probe (round 6 gate): the fused final conv's own cost -- the gate results are kept alive, nothing is folded or stored
    if constexpr (FINAL) {
#pragma unroll
        for (int nt = 0; nt < LN; ++nt)
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("" :: "v"(resid[nt][r]));
        (void)eps_acc; (void)ffs;
    }
#else
    if constexpr (FINAL) {
        // hop 256: utterance lengths are whole tiles, so every wave of a live workgroup is valid and reaches the barrier
        FD_STAMP_X(1);
        float *pb = reinterpret_cast<float *>(xs);                   // [part = 2 mt + hi][7 taps][256 columns]
        {
            const int part = 2 * mt0 + hi;
            float fw[8][8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float4 lo4 = ffs[(part * 8 + r) * 2], hi4 = ffs[(part * 8 + r) * 2 + 1];
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
        FD_STAMP_X(2);
        __syncthreads();
        FD_STAMP_X(3);
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
        FD_STAMP_X(4);
    }
#endif
    if (!(mx < GX_LIMIT)) {      // also inf; a NaN operand gives a NaN result on either path
        atomicOr(range_flag, 1);
        if constexpr (UP > 0) atomicOr(up_flag, 1);
    }
    FD_STAMP(7);
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
    const int b = blockIdx.y, w0 = lvc_tile_of_workgroup<(FD_LVC_XCD_RUN > 1 ? 2 * FD_LVC_XCD_RUN : 1)>() * W;      // runs of 16 tiles per XCD (fd_kernels_common.h)
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
                    ka[mt][hf][tap][0] = lvc_ld<128>(kp4 + 2 * e8);          // (bit 128: a hop-8 record has exactly one reader, this wave)
                    ka[mt][hf][tap][1] = lvc_ld<128>(kp4 + 2 * e8 + 1);
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

}  // namespace fdk_fast

// ------------------------------------------------------------------------------------------------
// stage drivers
// ------------------------------------------------------------------------------------------------
namespace fdk {
using namespace fdk_fast;

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
            FD_LAUNCH(L, HOP == 256 ? "lvc_up_h256" : "lvc_up_h64", (k_lvc_h2<HOP, 1, false, R>), dim3(((Ln + 255) / 256 + 7) / 8 * 8, B), dim3(256), 0, x_in, skip, x_out, kp, layer,
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
                if (c->final_fused)      // its own profile row: this variant never writes its 32 output channels
                    FD_LAUNCH(L, "lvc_final_h256", (k_lvc_h2<HOP, DIL, true>), dim3(((Ln + 255) / 256 + 7) / 8 * 8, B), dim3(256), 0, x_in, skip, x_out, kp,
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
        const dim3 grid(((Ln + 31) / 32 + 7) / 8 * 8, B);      // a multiple of 8: blockIdx.x % 8 is the XCD (lvc_tile_of_workgroup)
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

}  // namespace fdk
