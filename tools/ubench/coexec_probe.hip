// coexec_probe.hip -- do fp32 VALU instructions execute under a running fp32 MFMA (v_mfma_f32_32x32x2_f32) on gfx950?
// For each N: every MFMA of a dependent chain is followed by N independent v_fma_f32 (or v_mul+v_med3 pairs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef BF16_PIPE
#define MFLOP 32768.0
#else
#define MFLOP 4096.0
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NV, int KIND>
__global__ void __launch_bounds__(256, 2) probe(float *out, int iters, float seed)
{
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
    const float a = seed * 0.5f, b = seed * 0.25f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
#ifdef BF16_PIPE
            { bf16x8 av, bv; for (int q = 0; q < 8; ++q) { av[q] = (__bf16)a; bv[q] = (__bf16)b; }
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0); }
#else
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#endif
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) v[j & 15] = __builtin_fmaf(v[j & 15], a, b);
                else if (KIND == 1) v[j & 15] = __builtin_amdgcn_fmed3f(v[j & 15], v[j & 15] * 0.2f, 1e30f);
                else v[j & 15] = __builtin_amdgcn_exp2f(v[j & 15]);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + v[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NV, int KIND>
int run(float *out, int grid, const char *kind)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<NV, KIND>), dim3(grid), dim3(256), 0, 0, out, 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<NV, KIND>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)grid * 4 * iters * 32;
    // cycles per MFMA per SIMD assuming all waves of the grid are co-resident: waves/SIMD = grid*4/1024
    const double wps = grid * 4 / 1024.0;
    printf("%-8s NV=%2d grid=%4d (%.0f waves/SIMD): %7.3f ms  %6.1f TFLOP/s mfma  %5.1f ns per (MFMA + %d VALU) per wave\n", kind, NV, grid, wps, ms,
           mfma * MFLOP / ms / 1e9, ms * 1e6 / (iters * 32.0), NV);
    return 0;
}

int main()
{
    float *out;
    CK(hipMalloc(&out, 1 << 20));
    for (int grid : {256, 512}) {
        run<0, 0>(out, grid, "fma"); run<2, 0>(out, grid, "fma"); run<4, 0>(out, grid, "fma"); run<8, 0>(out, grid, "fma");
        run<16, 0>(out, grid, "fma"); run<32, 0>(out, grid, "fma");
        run<2, 1>(out, grid, "mul+med3"); run<8, 1>(out, grid, "mul+med3");
        run<2, 2>(out, grid, "exp2"); run<8, 2>(out, grid, "exp2");
    }
    return 0;
}
