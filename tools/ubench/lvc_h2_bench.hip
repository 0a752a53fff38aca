// lvc_h2_bench.hip -- standalone timing harness for k_lvc_h2 (phase stamps with s_memtime when FD_LVC_TIMING is set)
#include "../../fastdiff_amd/csrc/fd_kernels_lvc.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
bool fd_prof_stamps(const fdk::Launch &, const char *, hipEvent_t *, hipEvent_t *) { return false; }
void fd_prof_begin(const fdk::Launch &, const char *) {}
void fd_prof_end(const fdk::Launch &) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int HOP, int DIL>
int run(int B, int T, int reps)
{
    const int Ln = T * HOP;
    const size_t nx = (size_t)B * 32 * Ln, nk = (size_t)B * T * fd::KREC;
    float *x, *skip, *out, *kp, *wpack, *wref, *cb;
    int *flag;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&skip, nx * 4)); CK(hipMalloc(&out, nx * 4)); CK(hipMalloc(&kp, nk * 4));
    CK(hipMalloc(&wpack, 3072 * 4)); CK(hipMalloc(&wref, 3072 * 4)); CK(hipMalloc(&cb, 32 * 4)); CK(hipMalloc(&flag, 256)); CK(hipMemset(flag, 0, 256));
    std::vector<float> h(std::max(nx, nk));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CK(hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(skip, h.data(), nx * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nk; ++i) h[i] *= 0.05f;
    CK(hipMemcpy(kp, h.data(), nk * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wref, h.data(), 3072 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(cb, h.data(), 32 * 4, hipMemcpyHostToDevice));
    {   // fp16 weight pieces of plausible magnitude
        std::vector<unsigned short> w16(6144);
        unsigned s = 777u;
        for (auto &v : w16) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 16) & 0x8000u) | ((8u + ((s >> 8) % 6u)) << 10) | ((s >> 20) & 0x3FFu)); }
        CK(hipMemcpy(wpack, w16.data(), 6144 * 2, hipMemcpyHostToDevice));
    }
    dim3 grid(((Ln + 255) / 256 + 7) / 8 * 8, B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fdk_fast::k_lvc_h2<HOP, DIL, false>), grid, dim3(256), 0, 0, x, skip, out, kp, 1, (const float4 *)wpack, wref, cb, flag, T, (const int *)nullptr, (float *)nullptr, (const float4 *)nullptr, (const float4 *)nullptr, (const float *)nullptr, (int *)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fdk_fast::k_lvc_h2<HOP, DIL, false>), grid, dim3(256), 0, 0, x, skip, out, kp, 1, (const float4 *)wpack, wref, cb, flag, T, (const int *)nullptr, (float *)nullptr, (const float4 *)nullptr, (const float4 *)nullptr, (const float *)nullptr, (int *)nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double flops = 2.0 * B * T * HOP * (32 * 96 + 64 * 96), bytes = 4.0 * B * T * (96.0 * HOP + 6208);
    int hf = 0; CK(hipMemcpy(&hf, flag, 4, hipMemcpyDeviceToHost));
    printf("lvc_h2<%d,%d> B=%d T=%d: %.1f us  %.1f TFLOP/s(fp32-equivalent)  %.0f GB/s (algorithmic)  flag=%d\n", HOP, DIL, B, T, us, flops / us / 1e6, bytes / us / 1e3, hf);
#ifdef FD_LVC_TIMING
    std::vector<long long> d(64 * 4 * 8);
    CK(hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(fdk_fast::fd_dbg), d.size() * 8));
    double acc[8] = {0}, accw[4][8] = {{0}};
    for (int w = 0; w < 256; ++w) for (int i = 1; i < 8; ++i) { acc[i] += (double)(d[w * 8 + i] - d[w * 8 + i - 1]); accw[w % 4][i] += (double)(d[w * 8 + i] - d[w * 8 + i - 1]); }
    const char *names[8] = {"", "loads+stage+barrier", "resid + barrier", "conv MFMA + y split", "halo VALU", "barrier", "K split", "LVC + epilogue"};
    double tot = 0;
    for (int i = 1; i < 8; ++i) { printf("   %-22s %9.0f ticks   (wave0 %7.0f  wave1 %7.0f  wave2 %7.0f  wave3 %7.0f)\n", names[i], acc[i] / 256, accw[0][i] / 64, accw[1][i] / 64, accw[2][i] / 64, accw[3][i] / 64); tot += acc[i] / 256; }
    printf("   total %9.0f ticks per wave (s_memtime ticks)\n", tot);
#endif
    hipFree(x); hipFree(skip); hipFree(out); hipFree(kp);
    return 0;
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 864;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;      // (tools/power_trace.py runs ~2 s of the first configuration)
    run<256, 27>(B, T, reps);
    if (reps > 100) return 0;
    run<256, 1>(B, T, reps);
    run<64, 27>(B, T, reps);
    run<64, 1>(B, T, reps);
    return 0;
}
