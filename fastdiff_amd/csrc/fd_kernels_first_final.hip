// fd_kernels_first_final.hip -- a3 first_audio_conv, a10 final_conv + the reverse-step update (FastDiff_model.py:34-36,67-68,89,100; util.py:219-229)
// (one stage of the gfx950 kernel set; shared device helpers: fd_kernels_common.h; the one-thread-per-output twins: fd_kernels_naive.hip)
#include "fd_kernels_common.h"

namespace fdk_fast {

// =================================================================================================
// a3: first_audio_conv  Conv1d(1,32,k7,pad3)  (FastDiff_model.py:34-36,89)   -- VALU, HBM-write bound
// =================================================================================================
// advance != null (a sampler step that is not the first of its graph / launch sequence): the first workgroup does the previous step's
// end-of-step bookkeeping on the way -- next row of the step table, that step's range flags become "previous step" and join the call's
// sticky set -- what k_advance does in a launch of its own.  Nothing else in this kernel reads either; the kernels behind it start
// after the whole grid.
// The 224 weights + 32 biases reach the lanes through vector loads + LDS, not through scalar loads with a wave-uniform index: the scalar
// form was the one kernel property a short-lived neighbour process on the same compute units could disturb (round 4's two-process
// bisect: 86-95 mismatching calls per 16 400 with scalar loads, 0 per 75 810 with this form; LABBOOK.md, "two processes on one GPU").
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
    __shared__ float wl[fd::C * 8];      // [out][7 taps + bias]
    {
        const int o = threadIdx.x >> 3, k = threadIdx.x & 7;
        wl[threadIdx.x] = k < 7 ? w[o * 7 + k] : bias[o];
        __syncthreads();
    }
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int Lb = lens ? lens[b] * fd::HOPT : L;          // this utterance's own length (ragged batch)
    if (t0 >= Lb) return;
    const float *xr = x + (int64_t)b * L;
    float xv[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int p = t0 - 3 + i;
        xv[i] = (p >= 0 && p < Lb) ? xr[p] : 0.0f;
    }
#pragma unroll 4
    for (int o = 0; o < fd::C; ++o) {
        float wv[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) wv[k] = wl[o * 8 + k];
        const float bv = wl[o * 8 + 7];
        float4 r = make_float4(bv, bv, bv, bv);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            r.x += wv[k] * xv[k]; r.y += wv[k] * xv[k + 1]; r.z += wv[k] * xv[k + 2]; r.w += wv[k] * xv[k + 3];
        }
        lvc_st<64>(reinterpret_cast<float4 *>(a0 + ((int64_t)b * fd::C + o) * L + t0), r);
    }
}


// eps_acc (the final_conv sums of k_lvc_h2<..., FINAL>) -> eps = sum + bias -> eps_out or the reverse-step update; eps_acc is left
// zeroed for the next step.  If that LVC launch flagged its operands the sums are meaningless: they are only cleared here, and the
// plain k_final behind this launch (run_if) redoes the conv from the fp32 kernel's output.
__global__ void __launch_bounds__(256) k_final_acc(float *__restrict__ eps_acc, const float *__restrict__ bias, float *__restrict__ eps_out,
                                                   float *__restrict__ xstate, const StepParams *params, int sampler, int L,
                                                   const int *__restrict__ lens, const int *__restrict__ overflow)
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
        reinterpret_cast<float4 *>(xstate)[i4] = fdk::sampler_update4(xv, acc, params, b, (int64_t)(t0 >> 2));
    }
}

// =================================================================================================
// a10 + sampler: final_conv Conv1d(32,1,k7) (FastDiff_model.py:67-68,100) with the reverse-step update
// (util.py:219-229) fused into its epilogue.  VALU; each thread produces 4 consecutive samples.
// =================================================================================================
__global__ void __launch_bounds__(256) k_final(const float *__restrict__ x32, const float *__restrict__ w,
                                               const float *__restrict__ bias, float *__restrict__ eps_out,
                                               float *__restrict__ xstate, const StepParams *params, int sampler, int L,
                                               const int *__restrict__ lens, const int *__restrict__ run_if)
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
        reinterpret_cast<float4 *>(xstate)[i4] = fdk::sampler_update4(xv, acc, params, b, (int64_t)(t0 >> 2));
    }
}

}  // namespace fdk_fast

// ------------------------------------------------------------------------------------------------
// stage drivers
// ------------------------------------------------------------------------------------------------
namespace fdk {
using namespace fdk_fast;

hipError_t fast_first_conv(const Launch &L, const StepIO &io, int B, int T)
{
    const DevWeights &w = L.ctx->w;
    const int Lf = T * fd::HOPT;
    fd_context *c = L.ctx;
    const bool adv = io.sampler && c->advance_pending;      // the previous step of this sequence left its bookkeeping to us
    FD_LAUNCH(L, "first_conv", k_first_conv, dim3((Lf + 1023) / 1024, B), dim3(256), 0, io.x_in, w.first.w, w.first.b, c->ws.a[0], Lf, c->step_lens,
              adv ? c->ws.params : (StepParams *)nullptr, c->ws.range_flag);
    if (adv) c->advance_pending = false;
    return hipSuccess;
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
                  (const StepParams *)c->ws.params, io.sampler, Lf, c->step_lens, flag);
        run_if = flag;
        name = "final_conv_fallback";
        c->final_fused = false;
        if (!c->inline_fallback) return hipSuccess;      // fallback = host: a flagged last layer is redone from the host
    }
    FD_LAUNCH(L, name, k_final, dim3((Lf + 1023) / 1024, B), dim3(256), 0, x32, w.final_.w, w.final_.b, io.eps_out,
              c->ws.x, (const StepParams *)c->ws.params, io.sampler, Lf, c->step_lens, run_if);
    return hipSuccess;
}

}  // namespace fdk
