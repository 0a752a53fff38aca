// valu_rate_probe.hip -- issue cost (shader cycles per wave64 instruction) of the VALU / LDS instructions the LVC kernels are made of.
// Each kernel runs ITER x 64 copies of one instruction on 8 independent register sets (no dependency stalls), one workgroup of
// 256 threads per CU slot; cycles come from s_memtime of wave 0.  Run with 1 and 2 waves per SIMD (grid = #CU x 1 | 2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define DEFINE_PROBE(NAME, ASM)                                                                                          \
    __global__ void __launch_bounds__(256) NAME(long long *out, int iters, float seed)                                    \
    {                                                                                                                     \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;   \
        float b = seed * 0.5f + threadIdx.x, c = seed * 0.25f;                                                            \
        const float s = 1.5f;                                                                                             \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                                \
        for (int it = 0; it < iters; ++it) {                                                                              \
            REP64(ASM)                                                                                                    \
        }                                                                                                                 \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                                \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                  \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0;                                              \
    }

#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_FMA_ABS(i) asm volatile("v_fma_f32 %0, %0, %1, |%0|" : "+v"(a##i) : "s"(s));
#define A_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_MAX3(i) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a##i) : "v"(b), "v"(c));
#define A_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a##i) : "v"(b));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "s"(0x05040100));
#define A_CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_CVT32(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a##i));
#define A_MIX(i) asm volatile("v_fma_mix_f32 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(a##i) : "v"(b));
#define A_MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(a##i) : "v"(b), "s"(s));
#define A_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(q));
#define A_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
#define A_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
#define A_BITOP(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a##i) : "v"(b));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b));
#define A_FMAMK(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));

DEFINE_PROBE(p_fma, A_FMA)
DEFINE_PROBE(p_fma_abs, A_FMA_ABS)
DEFINE_PROBE(p_mul, A_MUL)
DEFINE_PROBE(p_max, A_MAX)
DEFINE_PROBE(p_max3, A_MAX3)
DEFINE_PROBE(p_med3, A_MED3)
DEFINE_PROBE(p_mov, A_MOV)
DEFINE_PROBE(p_perm, A_PERM)
DEFINE_PROBE(p_cvtpk, A_CVTPK)
DEFINE_PROBE(p_cvt32, A_CVT32)
DEFINE_PROBE(p_mix, A_MIX)
DEFINE_PROBE(p_mixlo, A_MIXLO)
DEFINE_PROBE(p_exp, A_EXP)
DEFINE_PROBE(p_rcp, A_RCP)
DEFINE_PROBE(p_xor, A_BITOP)
DEFINE_PROBE(p_lshlor, A_LSHLOR)
DEFINE_PROBE(p_cndmask, A_CNDMASK)
DEFINE_PROBE(p_fmac, A_FMAMK)

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) p_pkmul(long long *out, int iters, float seed)
{
    f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
    const f2 q = {seed * 0.5f + threadIdx.x, seed};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) { REP64(A_PKMUL) }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    const f2 r = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
    if (r.x + r.y == 12345.678f) out[0] = 0;
}
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##i) : "v"(q));
__global__ void __launch_bounds__(256) p_pkfma(long long *out, int iters, float seed)
{
    f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
    const f2 q = {seed * 0.5f + threadIdx.x, seed};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) { REP64(A_PKFMA) }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    const f2 r = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
    if (r.x + r.y == 12345.678f) out[0] = 0;
}

// LDS: ds_read_b128 of a conflict-free pattern, 8 independent destinations
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) p_dsread128(long long *out, int iters, float seed)
{
    __shared__ f4 buf[1024];
    buf[threadIdx.x] = f4{seed, seed, seed, seed};
    __syncthreads();
    f4 acc = {0, 0, 0, 0};
    const int base = threadIdx.x & 63;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            f4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base * 16), "n"((k & 7) * 1024));
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            acc += v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc.x + acc.y == 12345.678f) out[0] = 0;
}

template <typename K>
static int run(const char *name, K kern, int grid, long long *dout, int iters)
{
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dout, 4, 1.0f);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dout, iters, 1.0f);
    CK(hipDeviceSynchronize());
    long long h[8];
    CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-22s grid %4d: %6.2f cycles per instruction per wave (s_memtime ticks)\n", name, grid, (double)h[1] / (iters * 64.0));
    return 0;
}

int main()
{
    long long *dout;
    CK(hipMalloc(&dout, 8 * 4096));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 400;
    for (int mult : {1, 2}) {
        const int grid = cus * mult;
        printf("---- %d wave(s) per SIMD\n", mult);
        run("v_fma_f32", p_fma, grid, dout, iters);
        run("v_fma_f32 |abs| sgpr", p_fma_abs, grid, dout, iters);
        run("v_fmac_f32", p_fmac, grid, dout, iters);
        run("v_mul_f32", p_mul, grid, dout, iters);
        run("v_max_f32", p_max, grid, dout, iters);
        run("v_max3_f32", p_max3, grid, dout, iters);
        run("v_med3_f32", p_med3, grid, dout, iters);
        run("v_mov_b32", p_mov, grid, dout, iters);
        run("v_xor_b32", p_xor, grid, dout, iters);
        run("v_lshl_or_b32", p_lshlor, grid, dout, iters);
        run("v_cndmask_b32", p_cndmask, grid, dout, iters);
        run("v_perm_b32", p_perm, grid, dout, iters);
        run("v_cvt_pk_f16_f32", p_cvtpk, grid, dout, iters);
        run("v_cvt_f32_f16", p_cvt32, grid, dout, iters);
        run("v_fma_mix_f32", p_mix, grid, dout, iters);
        run("v_fma_mixlo_f16", p_mixlo, grid, dout, iters);
        run("v_pk_mul_f32", p_pkmul, grid, dout, iters);
        run("v_pk_fma_f32", p_pkfma, grid, dout, iters);
        run("v_exp_f32", p_exp, grid, dout, iters);
        run("v_rcp_f32", p_rcp, grid, dout, iters);
        run("ds_read_b128", p_dsread128, grid, dout, iters);
    }
    return 0;
}
