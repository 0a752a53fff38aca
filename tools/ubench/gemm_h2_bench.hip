// gemm_h2_bench.hip -- standalone timing harness for k_h_split + k_kp_gemm_h2 (build variants with -DFD_GX_NO_STORE / -DFD_GX_NO_FETCH)
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
    const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 864, G = argc > 3 ? atoi(argv[3]) : 512;
    const int R = fdk_fast::gx_rows(T);
    const size_t nh = (size_t)3 * B * 64 * T, nk = (size_t)3 * B * T * fd::KREC, ng = (size_t)776 * 2 * 12 * 64 * 4, nx = (size_t)3 * B * R * 64 + 1024;
    float *h, *kp, *g, *gb, *hx;
    int *flag;
    CK(hipMalloc(&flag, 256)); CK(hipMemset(flag, 0, 256));
    CK(hipMalloc(&h, nh * 4)); CK(hipMalloc(&kp, nk * 4)); CK(hipMalloc(&g, ng * 4)); CK(hipMalloc(&gb, fd::KREC * 4)); CK(hipMalloc(&hx, nx * 4));
    std::vector<float> v(ng);
    for (size_t i = 0; i < ng; ++i) v[i] = (float)((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CK(hipMemcpy(h, v.data(), nh * 4, hipMemcpyHostToDevice));
    {   // fp16 weight pieces of plausible magnitude (2^-7 .. 2^-2), random sign and mantissa
        std::vector<unsigned short> w16(ng * 2);
        unsigned x = 12345u;
        for (auto &b : w16) { x = x * 1664525u + 1013904223u; b = (unsigned short)(((x >> 16) & 0x8000u) | ((8u + ((x >> 8) % 6u)) << 10) | ((x >> 20) & 0x3FFu)); }
        CK(hipMemcpy(g, w16.data(), ng * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(gb, v.data(), fd::KREC * 4, hipMemcpyHostToDevice));
    const int tiles_per_utt = (T + 31) / 32, chunks = (tiles_per_utt + fdk_fast::GX_CT - 1) / fdk_fast::GX_CT;
    const int n_items = 3 * (fd::KREC / 128) * B * chunks;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&]() {
        hipLaunchKernelGGL(fdk_fast::k_h_split, dim3((32 * R + 255) / 256, 3 * B), dim3(256), 0, 0, h, (unsigned *)hx, flag, B, T, R, (const int *)nullptr);
        hipLaunchKernelGGL(fdk_fast::k_kp_gemm_h2, dim3(G), dim3(256), 0, 0, (const char *)hx, kp, (const float4 *)g, (const float4 *)g,
                           (const float4 *)g, gb, gb, gb, (const int *)flag, B, T, R, chunks, n_items, (const int *)nullptr, 0, 3);
    };
    for (int i = 0; i < 2; ++i) run();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int reps = argc > 4 ? atoi(argv[4]) : 5;      // (tools/power_trace.py runs ~2 s of it)
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, flops = 3 * 2.0 * 24832 * 192 * (double)B * T;
    printf("h_split+kp_gemm_x3 B=%d T=%d grid=%d items=%d: %.1f us  %.1f TFLOP/s(fp32-equivalent)\n", B, T, G, n_items, us, flops / us / 1e6);
#ifdef FD_GX_TIMING
    long long d[8];
    CK(hipMemcpyFromSymbol(d, HIP_SYMBOL(fdk_fast::fd_gxdbg), sizeof(d)));
    const char *nm[6] = {"weights check / loop", "DMA issue", "MFMA (2 tiles)", "stores issue (2 tiles)", "wait vmcnt", "barrier"};
    const double runs = 7.0 * G * 4, items = (double)n_items / G;
    for (int k = 0; k < 6; ++k) printf("   %-28s %7.1f ticks per item per wave\n", nm[k], d[k] / runs / items);
    printf("   whole kernel per wave: %.0f core ticks, %.1f us (s_memrealtime) -> %.0f MHz effective; loop share %.2f\n", d[6] / runs, d[7] / runs / 100.0,
           (double)d[6] / d[7] * 100.0, (double)(d[0] + d[1] + d[2] + d[3] + d[4] + d[5]) / d[6]);
#endif
    return 0;
}
