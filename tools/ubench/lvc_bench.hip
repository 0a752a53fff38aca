// lvc_bench.hip -- standalone timing harness for k_lvc_layer (phase stamps with s_memtime when FD_LVC_TIMING is set)
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
    using Cfg = fdk_fast::LvcCfg<HOP, DIL>;
    const int Ln = T * HOP;
    const size_t nx = (size_t)B * 32 * Ln, nk = (size_t)B * T * fd::KREC;
    float *x, *skip, *out, *kp, *wpack, *wref, *cb;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&skip, nx * 4)); CK(hipMalloc(&out, nx * 4)); CK(hipMalloc(&kp, nk * 4));
    CK(hipMalloc(&wpack, 3072 * 4)); CK(hipMalloc(&wref, 3072 * 4)); CK(hipMalloc(&cb, 32 * 4));
    std::vector<float> h(std::max(nx, nk));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CK(hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(skip, h.data(), nx * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nk; ++i) h[i] *= 0.05f;
    CK(hipMemcpy(kp, h.data(), nk * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wpack, h.data(), 3072 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wref, h.data(), 3072 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(cb, h.data(), 32 * 4, hipMemcpyHostToDevice));
    dim3 grid((Ln + Cfg::W - 1) / Cfg::W, B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fdk_fast::k_lvc_layer<HOP, DIL>), grid, dim3(256), 0, 0, x, skip, out, kp, 1, wpack, wref, cb, T, (const int *)nullptr, (const int *)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fdk_fast::k_lvc_layer<HOP, DIL>), grid, dim3(256), 0, 0, x, skip, out, kp, 1, wpack, wref, cb, T, (const int *)nullptr, (const int *)nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double flops = 2.0 * B * T * HOP * (32 * 96 + 64 * 96), bytes = 4.0 * B * T * (96.0 * HOP + 6208);
    printf("lvc_layer<%d,%d> B=%d T=%d: %.1f us  %.1f TFLOP/s  %.0f GB/s (algorithmic)\n", HOP, DIL, B, T, us, flops / us / 1e6, bytes / us / 1e3);
#ifdef FD_LVC_TIMING
    std::vector<long long> d(64 * 4 * 8);
    CK(hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(fdk_fast::fd_dbg), d.size() * 8));
    double acc[8] = {0};
    for (int w = 0; w < 256; ++w) for (int i = 1; i < 8; ++i) acc[i] += (double)(d[w * 8 + i] - d[w * 8 + i - 1]);
    const char *names[8] = {"", "Kload+stage+barrier", "conv MFMA + y write", "halo VALU", "barrier", "LVC mfma nt0", "epi0 + LVC mfma nt1", "epi1"};
    double tot = 0;
    for (int i = 1; i < 8; ++i) { printf("   %-22s %9.0f ticks\n", names[i], acc[i] / 256); tot += acc[i] / 256; }
    printf("   total %9.0f ticks per wave (s_memtime ticks)\n", tot);
    long long mn = d[0];
    for (int w = 0; w < 256; ++w) mn = std::min(mn, d[w * 8]);
    for (int w = 0; w < 256; w += 17) printf("      wg %3d wave %d: start %8lld  end %8lld  dur %6lld\n", w / 4, w % 4, d[w * 8] - mn, d[w * 8 + 7] - mn, d[w * 8 + 7] - d[w * 8]);
#endif
    hipFree(x); hipFree(skip); hipFree(out); hipFree(kp);
    return 0;
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 864;
    run<256, 27>(B, T, 20);
    run<256, 1>(B, T, 20);
    run<64, 27>(B, T, 20);
    run<8, 27>(B, T, 20);
    return 0;
}
