// fd_kernels_train.hip -- the location-variable convolution as a differentiable OPERATOR: forward and the three
// gradients of   out[b,o,q] = bias[b,o,l] + sum_{i,k} xpad[b,i,q + k - pad] * K[b,i,o,k,l],   l = q / hop,   pad = (ks - 1) / 2
// (modules/FastDiff/module/modules.py:220-253 with dilation = 1, the only value the model passes, modules.py:216), in the REFERENCE's
// own tensor layouts: x [B,Cin,L], K [B,Cin,Cout,ks,T] (T innermost), bias [B,Cout,T], out [B,Cout,L], L = T * hop.  This is what the
// reference's training step (theta_timestep_loss, util.py:291-325) differentiates through twelve times per forward; everything else
// of that step stays on PyTorch autograd around it (fastdiff_amd/train.py).
//
// Two kernel sets.  The model's own shape (Cin 32, Cout 64, ks 3, hop 8 / 64 / 256) runs on the exact-fp32 matrix instruction
// (v_mfma_f32_32x32x2_f32: training keeps fp32 products, no operand range to watch): per frame the operator is a small GEMM,
//   forward   Out_l [64 x hop] = K_l [64 x 96] Xwin_l [96 x hop]            dx   dX_l [32 x hop] = K_l' [32 x 192] dOutwin_l [192 x hop]
//   dK_l [64 x 96] = dOut_l [64 x hop] Xwin_l' [hop x 96]
// and the only awkward part is the reference's kernel layout, T innermost: a frame's 6144 coefficients lie T floats apart.  They are
// brought into frame-major order by a tiled transpose whose row order is the matrix instruction's A-operand order (one coalesced
// float4 per lane and four k-steps), the main kernels then stream them once; dK leaves its kernel frame-major in the accumulator
// layout and a second transpose puts it back.  The scratch buffer of B*T*6144 floats comes from the caller (fd_api.cpp).
// Any other shape takes the plain one-thread-per-output VALU kernels at the end of the file.
#include "fd_kernels.h"
#include "fd_frame_order.h"

namespace fdk_train {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float f4c(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
// D layout of the 32x32 tile: register r of lane (col = lane & 31, hi = lane >> 5) is row (r & 3) + 8 (r >> 2) + 4 hi
__device__ __forceinline__ int drow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

using namespace fdk_order;      // MI, MO, MK, ME, ORDER_*, row_of (fd_frame_order.h)

// K [B][6144 rows][T] -> frame-major [B][T][6144] in ORDER (64 x 64 tiles through LDS, both sides coalesced).  kbs: floats between two
// utterances of K (6144 T for a tensor of its own; 4 x that for one layer's slice of the predictor's [B, 4, 32, 64, 3, T] output)
template <int ORDER>
__global__ void __launch_bounds__(256) k_lvc_pack(const float *__restrict__ K, float *__restrict__ Kf, int T, int64_t kbs)
{
    __shared__ float tile[64][65];
    const int l0 = blockIdx.x * 64, e0 = blockIdx.y * 64, b = blockIdx.z, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int el = w * 16 + r, l = l0 + lane;
        tile[el][lane] = l < T ? K[(int64_t)b * kbs + (int64_t)row_of<ORDER>(e0 + el) * T + l] : 0.0f;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int ll = w * 16 + r, l = l0 + ll;
        if (l < T) Kf[((int64_t)b * T + l) * ME + e0 + lane] = tile[lane][ll];
    }
}

// frame-major dK (ORDER_DK) -> dK [B][6144 rows][T]
__global__ void __launch_bounds__(256) k_lvc_unpack(const float *__restrict__ Kf, float *__restrict__ K, int T, int64_t kbs)
{
    __shared__ float tile[64][65];
    const int l0 = blockIdx.x * 64, e0 = blockIdx.y * 64, b = blockIdx.z, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int ll = w * 16 + r, l = l0 + ll;
        tile[ll][lane] = l < T ? Kf[((int64_t)b * T + l) * ME + e0 + lane] : 0.0f;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int el = w * 16 + r, l = l0 + lane;
        if (l < T) K[(int64_t)b * kbs + (int64_t)row_of<ORDER_DK>(e0 + el) * T + l] = tile[lane][el];
    }
}

// frame-major ORDER_FWD (the "frames" entry points: what kernel_conv wrote) -> frame-major ORDER_DX for the dx kernel: a permutation
// inside each frame's 24 KB (gathered reads that stay in the L1 / L2 lines of the frame, coalesced writes).  kbs: floats between two
// utterances of the source.
__global__ void __launch_bounds__(256) k_lvc_reorder_dx(const float *__restrict__ Kf, float *__restrict__ Kx, int T, int64_t kbs)
{
    const int l = blockIdx.x, b = blockIdx.y;
    const float *src = Kf + (int64_t)b * kbs + (int64_t)l * ME;
    float *dst = Kx + ((int64_t)b * T + l) * ME;
    float v[ME / 256];
#pragma unroll
    for (int r = 0; r < ME / 256; ++r) {
        const int row = row_of<ORDER_DX>(r * 256 + threadIdx.x), k = row % MK, io = row / MK;
        v[r] = src[fwd_pos_of(io / MO, io % MO, k)];
    }
#pragma unroll
    for (int r = 0; r < ME / 256; ++r) dst[r * 256 + threadIdx.x] = v[r];
}

// Work split of the forward and dx kernels.  Workgroup = 256 threads = W columns: hop 256: one frame, hop 64: four frames, hop 8: four
// frames (32 columns).  Forward: hop 256: wave = (row tile, column half), 4 column tiles; hop 64 / 8: wave = frame, both row tiles.
// LDS rows: [3 pad][halo][W columns][halo][3 pad], so that the W columns are 16 B aligned (column c of the tile at index c + 4).
template <int HOP>
struct LvcGeo {
    static constexpr int W = HOP == 8 ? 32 : 256, FPW = W / HOP, LD = W + 8, C0 = 4;
};

// rows x (W + 2 halo) of a [rows][L] tensor into LDS, zero outside the signal: float4 for the aligned centre, scalars for the halo;
// every load is issued before the first LDS write
template <int ROWS, int W, int LD>
__device__ __forceinline__ void stage_rows(float *__restrict__ lds, const float *__restrict__ src, int L, int q0, int tid)
{
    constexpr int NF4 = ROWS * (W / 4), NV = (NF4 + 255) / 256, NH = (2 * ROWS + 255) / 256;
    float4 v[NV];
    float hv[NH];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = k * 256 + tid, r = idx / (W / 4), c4 = idx - r * (W / 4), q = q0 + 4 * c4;
        v[k] = (idx < NF4 && q < L) ? *reinterpret_cast<const float4 *>(src + (int64_t)r * L + q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
        const int idx = k * 256 + tid, r = idx >> 1, q = (idx & 1) ? q0 + W : q0 - 1;
        hv[k] = (idx < 2 * ROWS && q >= 0 && q < L) ? src[(int64_t)r * L + q] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = k * 256 + tid, r = idx / (W / 4), c4 = idx - r * (W / 4);
        if (idx < NF4) *reinterpret_cast<float4 *>(lds + r * LD + 4 + 4 * c4) = v[k];
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
        const int idx = k * 256 + tid, r = idx >> 1;
        if (idx < 2 * ROWS) lds[r * LD + ((idx & 1) ? 4 + W : 3)] = hv[k];
    }
}

template <int HOP>
__global__ void __launch_bounds__(256, HOP == 256 ? 3 : 2) k_lvc_fwd_mfma(const float *__restrict__ x, const float *__restrict__ Kf,
                                                                        const float *__restrict__ bias, float *__restrict__ out, int T,
                                                                        int64_t kfs, int64_t bbs)      // kfs / bbs: floats between two utterances of Kf / bias
{
    using G = LvcGeo<HOP>;
    constexpr int W = G::W, LD = G::LD, NMT = HOP == 256 ? 1 : 2, NCT = HOP == 256 ? 4 : (HOP == 64 ? 2 : 1);
    __shared__ __attribute__((aligned(16))) float xs[MI * LD];
    const int L = T * HOP, b = blockIdx.y, q0 = blockIdx.x * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, hi = lane >> 5;
    const int f = HOP == 256 ? blockIdx.x : blockIdx.x * G::FPW + wave;          // this wave's frame
    const int mt0 = HOP == 256 ? (wave & 1) : 0;
    const int cbase = HOP == 256 ? 128 * (wave >> 1) : HOP * wave;               // first column of the wave inside the tile
    const bool wave_valid = f < T;
    float4 a[NMT][12];
    if (wave_valid) {
        const float4 *ap = reinterpret_cast<const float4 *>(Kf + (int64_t)b * kfs + (int64_t)f * ME);
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int sq = 0; sq < 12; ++sq) a[m][sq] = ap[((mt0 + m) * 12 + sq) * 64 + lane];
    }
    stage_rows<MI, W, LD>(xs, x + (int64_t)b * MI * L, L, q0, tid);
    __syncthreads();
    if (!wave_valid) return;
    const int colc = HOP == 8 ? min(col, HOP - 1) : col;      // hop 8: a tile is one frame, 8 of its 32 columns exist
    float bz[NMT][16];
    {
        const float *bp = bias + (int64_t)b * bbs + (int64_t)(32 * mt0 + 4 * hi) * T + f;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) bz[m][r] = bp[(32 * m + (r & 3) + 8 * (r >> 2)) * T];
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        f32x16 acc[NMT];
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = bz[m][r];
        const int cl = G::C0 - 1 + cbase + 32 * ct + colc;    // x[q + tap - 1] = xs[.][C0 + (q - q0) + tap - 1]
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            const float v = xs[(2 * (s & 15) + hi) * LD + cl + (s >> 4)];
#pragma unroll
            for (int m = 0; m < NMT; ++m) acc[m] = mfma32(f4c(a[m][s >> 2], s & 3), v, acc[m]);
        }
        if (HOP != 8 || col < HOP) {
            const int q = q0 + cbase + 32 * ct + col;
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[((int64_t)b * MO + 32 * (mt0 + m) + drow(r, hi)) * L + q] = acc[m][r];
        }
    }
}

// dx.  Wave = 64 columns (hop 8: one frame), one 32-row tile, k = (tap, o).  The matrix part covers the taps that stay inside the
// column's own frame; a frame's first and last column also see the neighbour frames' kernels,
//   dx[i, l hop] += sum_o dout[o, l hop - 1] K[i, o, 2, l-1]          dx[i, (l+1) hop - 1] += sum_o dout[o, (l+1) hop] K[i, o, 0, l+1],
// 2 x 32 sums of 64 products per frame: the wave that owns the column computes them on VALU (lane = (i, parity of o): its eight
// float4 of the neighbour frame's operand copy are exactly those coefficients) and adds them before the store.
// FRAMES: Kx is not the ORDER_DX copy but the frames as kernel_conv wrote them (ORDER_FWD, kbs floats between utterances): coefficient
// (i, o, k) lies at (o >> 5) 3072 + k 1024 + (i >> 3) 256 + (i & 1) 128 + (o & 31) 4 + ((i >> 1) & 3) of its frame (fwd_pos_of), so a lane
// (i = col, parity hi of o) finds its 96 + 32 coefficients at one lane offset plus compile-time constants: dword gathers inside the
// frame's 24 KB (every line of it is used by some lane of the wave: L1 hits) instead of float4 loads of a transposed copy.
template <int HOP, bool FRAMES>
__global__ void __launch_bounds__(256, 2) k_lvc_dx_mfma(const float *__restrict__ dout, const float *__restrict__ Kx, float *__restrict__ dx,
                                                        int T, int64_t kbs)
{
    using G = LvcGeo<HOP>;
    constexpr int W = G::W, LD = G::LD, NCT = HOP == 8 ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float ds[MO * LD];
    const int L = T * HOP, b = blockIdx.y, q0 = blockIdx.x * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, hi = lane >> 5;
    const int f = HOP == 256 ? blockIdx.x : blockIdx.x * G::FPW + wave;
    const int cbase = HOP == 8 ? HOP * wave : 64 * wave;
    const int fbase = HOP == 256 ? 0 : HOP * wave;            // the frame's first column inside the tile
    const bool wave_valid = f < T;
    const bool own_first = HOP == 256 ? wave == 0 : true, own_last = HOP == 256 ? wave == 3 : true;
    float4 a[24], ef[8], el[8];
    const bool edge_f = wave_valid && own_first && f > 0, edge_l = wave_valid && own_last && f + 1 < T;
    if (wave_valid) {
        if constexpr (FRAMES) {
            const float *fp = Kx + (int64_t)b * kbs + (int64_t)f * ME + ((col >> 3) * 256 + (col & 1) * 128 + ((col >> 1) & 3) + 4 * hi);
            auto gather = [&](const float *frame, int sq) {      // the float4 of operand group sq: k index 2 (4 sq + j) + hi = tap * 64 + o
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int oe = (8 * sq + 2 * j) & 63, tap = (8 * sq + 2 * j) >> 6;
                    v[j] = frame[(oe >> 5) * 3072 + tap * 1024 + (oe & 31) * 4];
                }
                return make_float4(v[0], v[1], v[2], v[3]);
            };
#pragma unroll
            for (int sq = 0; sq < 24; ++sq) a[sq] = gather(fp, sq);
#pragma unroll
            for (int sq = 0; sq < 8; ++sq) {
                ef[sq] = edge_f ? gather(fp - ME, 16 + sq) : make_float4(0.f, 0.f, 0.f, 0.f);      // frame f-1, tap 2
                el[sq] = edge_l ? gather(fp + ME, sq) : make_float4(0.f, 0.f, 0.f, 0.f);           // frame f+1, tap 0
            }
        } else {
            const float4 *ap = reinterpret_cast<const float4 *>(Kx + ((int64_t)b * T + f) * ME);
#pragma unroll
            for (int sq = 0; sq < 24; ++sq) a[sq] = ap[sq * 64 + lane];
#pragma unroll
            for (int sq = 0; sq < 8; ++sq) {
                ef[sq] = edge_f ? (ap - ME / 4)[(16 + sq) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);      // frame f-1, tap 2
                el[sq] = edge_l ? (ap + ME / 4)[sq * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);             // frame f+1, tap 0
            }
        }
    }
    stage_rows<MO, W, LD>(ds, dout + (int64_t)b * MO * L, L, q0, tid);
    __syncthreads();
    if (!wave_valid) return;
    // lane (i = col, parity hi): o = 2 (4 sq + j) + hi
    float e0 = 0.0f, e1 = 0.0f;
    {
        const float *d0 = ds + G::C0 + fbase - 1, *d1 = ds + G::C0 + fbase + HOP;
#pragma unroll
        for (int sq = 0; sq < 8; ++sq)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 2 * (4 * sq + j) + hi;
                e0 = fmaf(f4c(ef[sq], j), d0[o * LD], e0);
                e1 = fmaf(f4c(el[sq], j), d1[o * LD], e1);
            }
        e0 += __shfl_xor(e0, 32, 64);
        e1 += __shfl_xor(e1, 32, 64);
    }
    const int colc = HOP == 8 ? min(col, HOP - 1) : col;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const int pl = cbase + 32 * ct + colc, pf = pl % HOP;       // column inside the tile / inside its frame
        const bool first = pf == 0, last = pf == HOP - 1;
        // dx[p] += dout[q] K[., ., tap], q = p - tap + 1 = ds[.][C0 + p - q0 + 1 - tap]; q must lie in p's own frame
#pragma unroll
        for (int s = 0; s < 96; ++s) {
            const int tap = s >> 5;
            float v = ds[(2 * (s & 31) + hi) * LD + G::C0 + pl + 1 - tap];
            if (tap == 0) v = last ? 0.0f : v;
            if (tap == 2) v = first ? 0.0f : v;
            acc = mfma32(f4c(a[s >> 2], s & 3), v, acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {      // (every lane takes part in the shuffles)
            const float g0 = __shfl(e0, drow(r, hi), 64), g1 = __shfl(e1, drow(r, hi), 64);
            acc[r] += (first ? g0 : 0.0f) + (last ? g1 : 0.0f);
        }
        if (HOP != 8 || col < HOP) {
            const int p = q0 + cbase + 32 * ct + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[((int64_t)b * MI + drow(r, hi)) * L + p] = acc[r];
        }
    }
}

// dK and dbias of one frame: 3 waves, wave = tap, both row tiles; the frame goes through LDS CH columns at a time (A: lanes along
// the rows of dout, B: lanes along the channels of x -- both row strides odd, so conflict-free), the next chunk's loads in flight
// under the current chunk's matrix work.  dK stays frame-major (ORDER_DK).
template <int CH>
__global__ void __launch_bounds__(192) k_lvc_dk_mfma(const float *__restrict__ x, const float *__restrict__ dout, float *__restrict__ dKf,
                                                     float *__restrict__ dbias, int T, int hop, int64_t dkfs,
                                                     int64_t dbbs)      // dkfs / dbbs: floats between two utterances of dKf / dbias
{
    constexpr int DLD = CH + 1, XLD = CH + 3, C4 = CH / 4;
    constexpr int ND = (MO * C4 + 191) / 192, NX = (MI * C4 + 191) / 192;
    __shared__ float ds[MO * DLD];
    __shared__ float xs[MI * XLD];
    __shared__ float red[3][MO];
    const int L = T * hop, l = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, tap = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const float *dp = dout + (int64_t)b * MO * L, *xp = x + (int64_t)b * MI * L;
    float4 dv[ND], xv[NX];
    float hv = 0.0f;
    auto load_chunk = [&](int qc) {
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = k * 192 + tid, o = idx / C4, c4 = idx - o * C4;
            dv[k] = idx < MO * C4 ? *reinterpret_cast<const float4 *>(dp + (int64_t)o * L + qc + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = k * 192 + tid, i = idx / C4, c4 = idx - i * C4;
            xv[k] = idx < MI * C4 ? *reinterpret_cast<const float4 *>(xp + (int64_t)i * L + qc + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < 2 * MI) {
            const int q = (tid & 1) ? qc + CH : qc - 1;
            hv = (q >= 0 && q < L) ? xp[(int64_t)(tid >> 1) * L + q] : 0.0f;
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    float accb = 0.0f;
    load_chunk(l * hop);
    for (int s0 = 0; s0 < hop; s0 += CH) {
        __syncthreads();                     // the previous chunk's reads are done
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = k * 192 + tid, o = idx / C4, c4 = idx - o * C4;
            if (idx < MO * C4) {
                float *w = ds + o * DLD + 4 * c4;
                w[0] = dv[k].x; w[1] = dv[k].y; w[2] = dv[k].z; w[3] = dv[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = k * 192 + tid, i = idx / C4, c4 = idx - i * C4;
            if (idx < MI * C4) {
                float *w = xs + i * XLD + 1 + 4 * c4;
                w[0] = xv[k].x; w[1] = xv[k].y; w[2] = xv[k].z; w[3] = xv[k].w;
            }
        }
        if (tid < 2 * MI) xs[(tid >> 1) * XLD + ((tid & 1) ? CH + 1 : 0)] = hv;
        __syncthreads();
        if (s0 + CH < hop) load_chunk(l * hop + s0 + CH);
#pragma unroll 8
        for (int st = 0; st < CH / 2; ++st) {
            const int sc = 2 * st + hi;
            const float bv = xs[l31 * XLD + sc + tap];                 // x[q + tap - 1]
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = mfma32(ds[(32 * m + l31) * DLD + sc], bv, acc[m]);
        }
        for (int c = tap; c < CH; c += 3) accb += ds[lane * DLD + c];   // thread = (row lane, every third column)
    }
    if (dKf) {
        float4 *dst = reinterpret_cast<float4 *>(dKf + (int64_t)b * dkfs + (int64_t)l * ME);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                dst[(((m * 3 + tap) * 4 + g) * 64) + lane] = make_float4(acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]);
    }
    if (dbias) {
        red[tap][lane] = accb;
        __syncthreads();
        if (tid < MO) dbias[(int64_t)b * dbbs + (int64_t)tid * T + l] = red[0][tid] + red[1][tid] + red[2][tid];
    }
}

// ---- any other shape: one thread per output ------------------------------------------------------------------------------------

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

// ---- the gate of an LVC layer with its residual (modules.py:217): out = x + sigmoid(y[:, :C]) * tanh(y[:, C:]) --------------------
// Under autograd the reference spends four elementwise kernels on it forward (sigmoid, tanh, mul, add) and eight backward, each
// moving the layer's whole [B, 32, L] tensor through HBM; here it is one pass each way: forward reads x and both halves of y,
// backward reads y and dout and writes both halves of dy (d out / d x is the identity: autograd hands dout on).
__device__ __forceinline__ float sigm(float a) { return 1.0f / (1.0f + expf(-a)); }
__device__ __forceinline__ float tanh_e(float b) { return 1.0f - 2.0f / (expf(2.0f * b) + 1.0f); }      // exact limits at +-inf

template <int VEC>
__global__ void __launch_bounds__(256) k_gate_fwd(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out, int C,
                                                  int64_t L)
{
    const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (q >= L) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const float *ya = y + ((int64_t)b * 2 * C + c) * L + q, *yb = ya + (int64_t)C * L;
    const int64_t xo = ((int64_t)b * C + c) * L + q;
    if (VEC == 4) {
        const float4 a = *reinterpret_cast<const float4 *>(ya), t = *reinterpret_cast<const float4 *>(yb), xv = *reinterpret_cast<const float4 *>(x + xo);
        *reinterpret_cast<float4 *>(out + xo) = make_float4(xv.x + sigm(a.x) * tanh_e(t.x), xv.y + sigm(a.y) * tanh_e(t.y),
                                                            xv.z + sigm(a.z) * tanh_e(t.z), xv.w + sigm(a.w) * tanh_e(t.w));
    } else {
        out[xo] = x[xo] + sigm(ya[0]) * tanh_e(yb[0]);
    }
}

template <int VEC>
__global__ void __launch_bounds__(256) k_gate_bwd(const float *__restrict__ y, const float *__restrict__ dout, float *__restrict__ dy, int C,
                                                  int64_t L)
{
    const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (q >= L) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t ao = ((int64_t)b * 2 * C + c) * L + q, bo = ao + (int64_t)C * L, xo = ((int64_t)b * C + c) * L + q;
    auto da = [](float g, float a, float t) { const float s = sigm(a), th = tanh_e(t); return g * th * s * (1.0f - s); };
    auto db = [](float g, float a, float t) { const float s = sigm(a), th = tanh_e(t); return g * s * (1.0f - th * th); };
    if (VEC == 4) {
        const float4 a = *reinterpret_cast<const float4 *>(y + ao), t = *reinterpret_cast<const float4 *>(y + bo), g = *reinterpret_cast<const float4 *>(dout + xo);
        *reinterpret_cast<float4 *>(dy + ao) = make_float4(da(g.x, a.x, t.x), da(g.y, a.y, t.y), da(g.z, a.z, t.z), da(g.w, a.w, t.w));
        *reinterpret_cast<float4 *>(dy + bo) = make_float4(db(g.x, a.x, t.x), db(g.y, a.y, t.y), db(g.z, a.z, t.z), db(g.w, a.w, t.w));
    } else {
        dy[ao] = da(dout[xo], y[ao], y[bo]);
        dy[bo] = db(dout[xo], y[ao], y[bo]);
    }
}

// ---- a skip tensor's fan-out (FastDiff_model.py:91-98): x feeds the DiffusionDBlock below it -- which begins by picking every f-th
//      column (F.interpolate, nearest: modules.py:128-131) -- and, as audio_down, the four layers of the LVC block at its rate
//      (modules.py:209).  Forward: the pick.  Backward: dx = g0 + g1 + g2 + g3 + scatter(gp), one pass instead of a zero-fill, a
//      strided scatter and four full-size additions under autograd.  rows = B * C; any gradient pointer may be null.
__global__ void __launch_bounds__(256) k_fan_pick(const float *__restrict__ x, float *__restrict__ out, int64_t L, int64_t Lo, int f)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < Lo) out[(int64_t)blockIdx.y * Lo + j] = x[(int64_t)blockIdx.y * L + j * f];
}

template <int VEC>
__global__ void __launch_bounds__(256) k_fan_sum(const float *__restrict__ g0, const float *__restrict__ g1, const float *__restrict__ g2,
                                                 const float *__restrict__ g3, const float *__restrict__ gp, float *__restrict__ dx, int64_t L,
                                                 int64_t Lo, int f)
{
    const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (q >= L) return;
    const int64_t at = (int64_t)blockIdx.y * L + q;
    if (VEC == 4) {      // (f is a multiple of 4: column q is the only one of the quad that may have been picked)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        auto add = [&](const float *g) { if (g) { const float4 t = *reinterpret_cast<const float4 *>(g + at); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; } };
        add(g0); add(g1); add(g2); add(g3);
        if (gp && q % f == 0 && q / f < Lo) v.x += gp[(int64_t)blockIdx.y * Lo + q / f];
        *reinterpret_cast<float4 *>(dx + at) = v;
    } else {
        float v = 0.0f;
        if (g0) v += g0[at];
        if (g1) v += g1[at];
        if (g2) v += g2[at];
        if (g3) v += g3[at];
        if (gp && q % f == 0 && q / f < Lo) v += gp[(int64_t)blockIdx.y * Lo + q / f];
        dx[at] = v;
    }
}

}  // namespace fdk_train

namespace fdk {
using namespace fdk_train;

hipError_t fan_pick(const Launch &L, const float *x, float *out, int rows, int64_t len, int f)
{
    const int64_t Lo = len / f;
    FD_LAUNCH(L, "fan_pick", k_fan_pick, dim3((unsigned)((Lo + 255) / 256), rows), dim3(256), 0, x, out, len, Lo, f);
    return hipSuccess;
}

hipError_t fan_sum(const Launch &L, const float *const g[4], const float *gp, float *dx, int rows, int64_t len, int f)
{
    const int64_t Lo = len / f;
    if (len % 4 == 0 && f % 4 == 0)
        FD_LAUNCH(L, "fan_sum", k_fan_sum<4>, dim3((unsigned)((len / 4 + 255) / 256), rows), dim3(256), 0, g[0], g[1], g[2], g[3], gp, dx, len, Lo, f);
    else
        FD_LAUNCH(L, "fan_sum", k_fan_sum<1>, dim3((unsigned)((len + 255) / 256), rows), dim3(256), 0, g[0], g[1], g[2], g[3], gp, dx, len, Lo, f);
    return hipSuccess;
}

hipError_t gate_forward(const Launch &L, const float *x, const float *y, float *out, int B, int C, int64_t len)
{
    if (len % 4 == 0) FD_LAUNCH(L, "gate_forward", k_gate_fwd<4>, dim3((unsigned)((len / 4 + 255) / 256), C, B), dim3(256), 0, x, y, out, C, len);
    else FD_LAUNCH(L, "gate_forward", k_gate_fwd<1>, dim3((unsigned)((len + 255) / 256), C, B), dim3(256), 0, x, y, out, C, len);
    return hipSuccess;
}

hipError_t gate_backward(const Launch &L, const float *y, const float *dout, float *dy, int B, int C, int64_t len)
{
    if (len % 4 == 0) FD_LAUNCH(L, "gate_backward", k_gate_bwd<4>, dim3((unsigned)((len / 4 + 255) / 256), C, B), dim3(256), 0, y, dout, dy, C, len);
    else FD_LAUNCH(L, "gate_backward", k_gate_bwd<1>, dim3((unsigned)((len + 255) / 256), C, B), dim3(256), 0, y, dout, dy, C, len);
    return hipSuccess;
}

static bool model_shape(int Cin, int Cout, int ks, int hop) { return Cin == MI && Cout == MO && ks == MK && (hop == 8 || hop == 64 || hop == 256); }
bool lvc_op_needs_scratch(int Cin, int Cout, int ks, int hop) { return model_shape(Cin, Cout, ks, hop); }

// frames = true (the model's shape only): K is frame-major ORDER_FWD ([T][6144] per utterance, kbs floats between utterances: what
// kconv_forward_frames wrote), dK leaves frame-major ORDER_DK (dkbs between utterances: what kconv_backward_frames reads) -- no
// transposes; the scratch then only holds the ORDER_DX copy for the dx kernel
static hipError_t launch_fwd_mfma(const Launch &L, const float *x, const float *Kf, const float *bias, float *out, int B, int T, int hop, int64_t kfs,
                                  int64_t bbs)
{
    if (bbs == 0) bbs = (int64_t)MO * T;
    if (hop == 256) FD_LAUNCH(L, "lvc_op_forward", k_lvc_fwd_mfma<256>, dim3(T, B), dim3(256), 0, x, Kf, bias, out, T, kfs, bbs);
    else if (hop == 64) FD_LAUNCH(L, "lvc_op_forward", k_lvc_fwd_mfma<64>, dim3((T + 3) / 4, B), dim3(256), 0, x, Kf, bias, out, T, kfs, bbs);
    else FD_LAUNCH(L, "lvc_op_forward", k_lvc_fwd_mfma<8>, dim3((T + 3) / 4, B), dim3(256), 0, x, Kf, bias, out, T, kfs, bbs);
    return hipSuccess;
}

hipError_t lvc_op_forward(const Launch &L, const float *x, const float *K, const float *bias, float *out, int B, int Cin, int Cout, int ks,
                          int T, int hop, float *scratch, int64_t kbs, bool frames, int64_t bbs)
{
    const int Ln = T * hop;
    if (kbs == 0) kbs = (int64_t)Cin * Cout * ks * T;
    if (frames) {
        if (!model_shape(Cin, Cout, ks, hop)) return hipErrorInvalidValue;
        return launch_fwd_mfma(L, x, K, bias, out, B, T, hop, kbs, bbs);
    }
    if (bbs != 0 && bbs != (int64_t)Cout * T) return hipErrorInvalidValue;      // (a batch-strided bias: the frames entry points only)
    if (!scratch || !model_shape(Cin, Cout, ks, hop)) {
        if (kbs != (int64_t)Cin * Cout * ks * T) return hipErrorInvalidValue;      // the generic kernels take a tensor of its own only
        FD_LAUNCH(L, "lvc_op_forward", k_lvc_fwd, dim3((Ln + 255) / 256, Cout, B), dim3(256), 0, x, K, bias, out, Cin, Cout, ks, T, hop);
        return hipSuccess;
    }
    FD_LAUNCH(L, "lvc_op_pack", k_lvc_pack<ORDER_FWD>, dim3((T + 63) / 64, ME / 64, B), dim3(256), 0, K, scratch, T, kbs);
    return launch_fwd_mfma(L, x, scratch, bias, out, B, T, hop, (int64_t)T * ME, 0);
}

hipError_t lvc_op_backward(const Launch &L, const float *x, const float *K, const float *dout, float *dx, float *dK, float *dbias, int B,
                           int Cin, int Cout, int ks, int T, int hop, float *scratch, int64_t kbs, int64_t dkbs, bool frames, int64_t dbbs)
{
    if (dbbs != 0 && dbbs != (int64_t)Cout * T && !frames) return hipErrorInvalidValue;      // (a batch-strided dbias: the frames entry points only)
    if (dbbs == 0) dbbs = (int64_t)Cout * T;
    const int Ln = T * hop;
    const int64_t own = (int64_t)Cin * Cout * ks * T;
    if (kbs == 0) kbs = own;
    if (dkbs == 0) dkbs = own;
    if (frames && (!model_shape(Cin, Cout, ks, hop) || (dx && !L.ctx->lvc_dx_gather && !scratch))) return hipErrorInvalidValue;
    if (!frames && (!scratch || !model_shape(Cin, Cout, ks, hop))) {
        if (kbs != own || dkbs != own) return hipErrorInvalidValue;
        if (dx) FD_LAUNCH(L, "lvc_op_backward_x", k_lvc_bwd_x, dim3((Ln + 255) / 256, Cin, B), dim3(256), 0, dout, K, dx, Cin, Cout, ks, T, hop);
        if (dK || dbias) {
            const size_t shmem = sizeof(float) * ((size_t)Cout * DK_CHUNK + (size_t)Cin * (DK_CHUNK + ks - 1));
            if (shmem > 48 * 1024) {      // wide outputs (Cout up to 256 is admitted) ask for more dynamic LDS than a launch gets by default
                const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(k_lvc_bwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                if (ea != hipSuccess) return ea;
            }
            FD_LAUNCH(L, "lvc_op_backward_k", k_lvc_bwd_k, dim3(T, B), dim3(256), shmem, x, dout, dK, dbias, Cin, Cout, ks, T, hop);
        }
        return hipSuccess;
    }
    if (dx && frames && L.ctx->lvc_dx_gather) {      // the dx kernel reads kernel_conv's frames where they lie
        if (hop == 256) FD_LAUNCH(L, "lvc_op_backward_x", (k_lvc_dx_mfma<256, true>), dim3(T, B), dim3(256), 0, dout, K, dx, T, kbs);
        else if (hop == 64) FD_LAUNCH(L, "lvc_op_backward_x", (k_lvc_dx_mfma<64, true>), dim3((T + 3) / 4, B), dim3(256), 0, dout, K, dx, T, kbs);
        else FD_LAUNCH(L, "lvc_op_backward_x", (k_lvc_dx_mfma<8, true>), dim3((T + 3) / 4, B), dim3(256), 0, dout, K, dx, T, kbs);
    } else if (dx) {
        if (frames) FD_LAUNCH(L, "lvc_op_reorder", k_lvc_reorder_dx, dim3(T, B), dim3(256), 0, K, scratch, T, kbs);
        else FD_LAUNCH(L, "lvc_op_pack", k_lvc_pack<ORDER_DX>, dim3((T + 63) / 64, ME / 64, B), dim3(256), 0, K, scratch, T, kbs);
        const int64_t own_fs = (int64_t)T * ME;
        if (hop == 256) FD_LAUNCH(L, "lvc_op_backward_x", (k_lvc_dx_mfma<256, false>), dim3(T, B), dim3(256), 0, dout, scratch, dx, T, own_fs);
        else if (hop == 64) FD_LAUNCH(L, "lvc_op_backward_x", (k_lvc_dx_mfma<64, false>), dim3((T + 3) / 4, B), dim3(256), 0, dout, scratch, dx, T, own_fs);
        else FD_LAUNCH(L, "lvc_op_backward_x", (k_lvc_dx_mfma<8, false>), dim3((T + 3) / 4, B), dim3(256), 0, dout, scratch, dx, T, own_fs);
    }
    if (dK || dbias) {
        // (the dx kernels are done with the scratch: same stream)
        float *dKf = !dK ? nullptr : frames ? dK : scratch;
        const int64_t dkfs = frames ? dkbs : (int64_t)T * ME;
        if (hop == 8) FD_LAUNCH(L, "lvc_op_backward_k", k_lvc_dk_mfma<8>, dim3(T, B), dim3(192), 0, x, dout, dKf, dbias, T, hop, dkfs, dbbs);
        else FD_LAUNCH(L, "lvc_op_backward_k", k_lvc_dk_mfma<64>, dim3(T, B), dim3(192), 0, x, dout, dKf, dbias, T, hop, dkfs, dbbs);
        if (dK && !frames) FD_LAUNCH(L, "lvc_op_unpack", k_lvc_unpack, dim3((T + 63) / 64, ME / 64, B), dim3(256), 0, scratch, dK, T, dkbs);
    }
    return hipSuccess;
}

}  // namespace fdk
