// clk_probe.hip -- what does s_memtime count, and what clock does a busy SIMD run at?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k_valu(long long *out, int n, float s)
{
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = s;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) a = a * 1.0001f + 0.5f;       // 64 dependent v_fma per iteration
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (a == 12345.0f) out[2] = 1;
}
__global__ void k_mfma(long long *out, int n, float s)
{
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(s + q); b[q] = (_Float16)(s - q); }
    f32x16 acc = {0};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (acc[0] == 12345.0f) out[2] = 1;
}
__global__ void k_mfma_rand(long long *out, int n, float s)
{
    // two operand sets with lane-dependent pseudo-random bit patterns, alternated: every MFMA sees fresh data
    const unsigned l = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    f16x8 a1, b1, a2, b2;
    for (int q = 0; q < 8; ++q) {
        a1[q] = (_Float16)(s * (float)((l >> (q + 3)) & 1023u) * 0.001f - 0.5f);
        b1[q] = (_Float16)(s * (float)((l >> (q + 7)) & 1023u) * 0.001f - 0.5f);
        a2[q] = (_Float16)(s * (float)((l >> (q + 11)) & 1023u) * 0.001f - 0.5f);
        b2[q] = (_Float16)(s * (float)((l >> (q + 13)) & 1023u) * 0.001f - 0.5f);
    }
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc = {0};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (acc[0] == 12345.0f) out[2] = 1;
}
int main()
{
    long long *d, h[3];
    hipMalloc(&d, 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        const int grid = (mode & 1) ? 2048 : 1, n = 20000;
        float ms;
        hipEventRecord(e0, 0);
        if (mode < 2) hipLaunchKernelGGL(k_valu, dim3(grid), dim3(256), 0, 0, d, n, 1.0f);
        else hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, d, n, 1.0f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        const double ops = (mode < 2) ? 64.0 * n : 32.0 * n;
        printf("%s grid %4d: wall %.3f ms | s_memtime %lld ticks (%.2f per op) | s_memrealtime %lld ticks -> %.1f MHz realtime, memtime %.1f MHz\n",
               mode < 2 ? "valu chain" : "mfma chain", grid, ms, h[0], h[0] / ops, h[1], h[1] / (ms * 1e3), h[0] / (ms * 1e3));
    }
    for (int grid = 1; grid <= 2048; grid *= 2048) {
        const int n = 20000;
        hipLaunchKernelGGL(k_mfma_rand, dim3(grid), dim3(256), 0, 0, d, n, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("mfma chain, alternating random operands, grid %4d: block 0 ran at %.0f MHz (%lld core ticks / %lld x 10 ns)\n", grid, (double)h[0] / h[1] * 100.0, h[0], h[1]);
    }
    for (int grid = 1; grid <= 2048; grid *= 2048) {
        hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, d, 20000, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("mfma chain, constant operands,           grid %4d: block 0 ran at %.0f MHz\n", grid, (double)h[0] / h[1] * 100.0);
    }
    return 0;
}
