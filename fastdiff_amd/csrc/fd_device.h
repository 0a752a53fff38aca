// fd_device.h -- device helpers shared by the kernel files (header-only: no relocatable device code needed).
#pragma once
#include "fd_internal.h"

namespace fdk {

// ---- Philox4x32-10 + Box-Muller -----------------------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Four N(0,1) draws: key = seed, counter = (float4 index, utterance stream id, step stream).  Without per-utterance ids (uid = 0,
// idx4 = the flat index into the batch) a draw depends on where the utterance sits in the batch; with them (fd_set_noise_streams:
// idx4 = the index inside the utterance, uid = its id) an utterance gets the same noise however it is batched or sharded.
__device__ inline float4 philox_normal4(unsigned long long seed, uint32_t stream, uint64_t idx4, unsigned long long uid = 0ull)
{
    uint32_t r[4];
    philox4x32_10((uint32_t)idx4, (uint32_t)(idx4 >> 32) ^ (uint32_t)uid, stream, 0x5EEDu ^ (uint32_t)(uid >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u0 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0);
    sincosf(6.283185307179586f * u3, &s1, &c1);
    return make_float4(ra * c0, ra * s0, rb * c1, rb * s1);
}

// The reverse-step update (util.py:219-229) on 4 consecutive samples; eps in, x in/out.  (b, off4) = utterance and float4 offset inside
// it.  The library's own buffers may be padded to a frame bucket (StepParams::l4 float4s per utterance); whatever belongs to the CALLER
// -- injected noise, the returned sequence, the position a Philox draw is keyed on -- is addressed with the caller's own length
// (l4_io per utterance, n4_io per step), so a bucketed call reads, draws and writes exactly what the unpadded call does.
__device__ inline float4 sampler_update4(float4 x, float4 e, const StepParams *p, int b, int64_t off4)
{
    const int64_t i4 = (int64_t)b * p->l4_io + off4, n4_total = p->n4_io;
    const int k = p->step_idx;
    const fd_step st = p->table[k];
    float4 o;
    if (p->ddim) {   // x = c1*x + c2*eps + c3*eps, evaluated left to right
        o.x = (st.c1 * x.x + st.c2 * e.x) + st.c3 * e.x;
        o.y = (st.c1 * x.y + st.c2 * e.y) + st.c3 * e.y;
        o.z = (st.c1 * x.z + st.c2 * e.z) + st.c3 * e.z;
        o.w = (st.c1 * x.w + st.c2 * e.w) + st.c3 * e.w;
    } else {         // x -= c_eps*eps ; x /= c_div ; if n>0: x = x + sigma*z
        o.x = (x.x - st.c_eps * e.x) / st.c_div;
        o.y = (x.y - st.c_eps * e.y) / st.c_div;
        o.z = (x.z - st.c_eps * e.z) / st.c_div;
        o.w = (x.w - st.c_eps * e.w) / st.c_div;
        if (st.add_noise) {
            float4 z;
            if (p->z) z = reinterpret_cast<const float4 *>(p->z)[(int64_t)k * n4_total + i4];
            else if (p->uids) z = philox_normal4(p->seed, (uint32_t)k, (uint64_t)off4, p->uids[b]);
            else z = philox_normal4(p->seed, (uint32_t)k, (uint64_t)i4);
            o.x += st.sigma * z.x; o.y += st.sigma * z.y; o.z += st.sigma * z.z; o.w += st.sigma * z.w;
        }
    }
    if (p->seq) reinterpret_cast<float4 *>(p->seq)[(int64_t)(k + 1) * n4_total + i4] = o;
    return o;
}


}  // namespace fdk
