// fd_frame_order.h -- internal: the frame-major orders of one frame's 6144 predicted-kernel coefficients (training path).
// Position e of a frame's 6144 floats <-> row (i * 64 + o) * 3 + k of the reference layout [B, Cin = 32, Cout = 64, ks = 3, T]
// (modules/FastDiff/module/modules.py:333-338).  e = (group * 64 + lane) * 4 + j: the float4 a lane loads covers four consecutive
// k-steps (j) of one 32x32x2 matrix-instruction operand column.  Shared by the LVC operator's kernels (fd_kernels_train.hip), which
// consume / produce these orders, and the predictor's kernel_conv (fd_kernels_kconv.hip), which can write the first and read the
// last directly (the "frames" entry points: no transposes between the two).
#pragma once

namespace fdk_order {

constexpr int MI = 32, MO = 64, MK = 3, ME = MI * MO * MK;      // the model's operator: 6144 coefficients per frame

enum { ORDER_FWD = 0, ORDER_DX = 1, ORDER_DK = 2 };

template <int ORDER>
__host__ __device__ __forceinline__ int row_of(int e)
{
    const int j = e & 3, lane = (e >> 2) & 63, grp = e >> 8, l31 = lane & 31, hi = lane >> 5;
    int i, o, k;
    if (ORDER == ORDER_FWD) {          // A[o][tap*32 + i]: grp = mt*12 + sq, k index = 2 (4 sq + j) + hi
        const int mt = grp / 12, sq = grp - 12 * mt, kidx = 2 * (4 * sq + j) + hi;
        o = 32 * mt + l31; k = kidx >> 5; i = kidx & 31;
    } else if (ORDER == ORDER_DX) {    // A[i][tap*64 + o]: grp = sq (24), k index = 2 (4 sq + j) + hi
        const int kidx = 2 * (4 * grp + j) + hi;
        i = l31; k = kidx >> 6; o = kidx & 63;
    } else {                           // the dK accumulators: grp = ((mt*3 + tap)*4 + g), rows 32 mt + 8 g + 4 hi + j, column i
        const int g = grp & 3, t6 = grp >> 2, mt = t6 / 3;
        k = t6 - 3 * mt; o = 32 * mt + 8 * g + 4 * hi + j; i = l31;
    }
    return (i * MO + o) * MK + k;
}

// where row (i * 64 + o) * 3 + k lies in a frame written in ORDER_FWD
__host__ __device__ __forceinline__ int fwd_pos_of(int i, int o, int k)
{
    const int kidx = k * 32 + i, hi = kidx & 1, q = kidx >> 1;      // q = 4 sq + j
    return ((((o >> 5) * 12 + (q >> 2)) * 64 + hi * 32 + (o & 31)) << 2) + (q & 3);
}

}  // namespace fdk_order
