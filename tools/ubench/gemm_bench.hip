// gemm_bench.hip -- standalone timing harness for k_kp_gemm
#include "../../fastdiff_amd/csrc/fd_kernels_kp.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
bool fd_prof_stamps(const fdk::Launch &, const char *, hipEvent_t *, hipEvent_t *) { return false; }
void fd_prof_begin(const fdk::Launch &, const char *) {}
void fd_prof_end(const fdk::Launch &) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 864;
    const size_t nh = (size_t)3 * B * 64 * T, nk = (size_t)3 * B * T * fd::KREC, ng = (size_t)776 * 24 * 256;
    float *h, *kp, *g, *gb;
    CK(hipMalloc(&h, nh * 4)); CK(hipMalloc(&kp, nk * 4)); CK(hipMalloc(&g, ng * 4)); CK(hipMalloc(&gb, fd::KREC * 4));
    std::vector<float> v(ng);
    for (size_t i = 0; i < ng; ++i) v[i] = (float)((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CK(hipMemcpy(g, v.data(), ng * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(h, v.data(), nh * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gb, v.data(), fd::KREC * 4, hipMemcpyHostToDevice));
    const int tiles_per_utt = (T + 31) / 32, chunks = (tiles_per_utt + fdk_fast::GEMM_CT - 1) / fdk_fast::GEMM_CT;
    const int chunk_tiles = (tiles_per_utt + chunks - 1) / chunks;
    const int n_items = 3 * (fd::KREC / 128) * B * chunks;
    dim3 grid(512, 1, 1);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fdk_fast::k_kp_gemm, grid, dim3(256), 0, 0, h, kp, g, g, g, gb, gb, gb, B, T, chunks, chunk_tiles, n_items, (const int *)nullptr, (const int *)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fdk_fast::k_kp_gemm, grid, dim3(256), 0, 0, h, kp, g, g, g, gb, gb, gb, B, T, chunks, chunk_tiles, n_items, (const int *)nullptr, (const int *)nullptr);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, flops = 3 * 2.0 * 24832 * 192 * (double)B * T;
    printf("kp_gemm B=%d T=%d chunk_tiles=%d grid=(%d,%d): %.1f us  %.1f TFLOP/s\n", B, T, chunk_tiles, grid.x, grid.y, us, flops / us / 1e6);
#ifdef FD_GEMM_TIMING
    std::vector<long long> d(64 * 4 * 4);
    CK(hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(fdk_fast::fd_gdbg), d.size() * 8));
    double a[4] = {0};
    for (int w = 0; w < 256; ++w) for (int i = 1; i < 4; ++i) a[i] += (double)(d[w * 4 + i] - d[w * 4 + i - 1]);
    printf("   prologue (weights + stage + barrier) %8.0f\n   first tile %8.0f\n   remaining %d tiles %8.0f (%.0f per tile)\n", a[1] / 256, a[2] / 256, chunk_tiles - 1, a[3] / 256, a[3] / 256 / (chunk_tiles - 1));
#endif
    return 0;
}
