// lvc_h8_bench.hip -- standalone timing harness for the hop-8 layer: k_lvc_h8m (16x16x32 fp16 matrix tiles) and k_lvc_h8 (all VALU, fp32).  Repetitions re-read the same 172 MB of records, which then sit in the
// 256 MB memory-side cache: the number is the kernel's own ceiling (42 us, 4.5 TB/s algorithmic), not what it sees behind the GEMM.
#include "../../fastdiff_amd/csrc/fd_kernels_lvc.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
bool fd_prof_stamps(const fdk::Launch &, const char *, hipEvent_t *, hipEvent_t *) { return false; }
void fd_prof_begin(const fdk::Launch &, const char *) {}
void fd_prof_end(const fdk::Launch &) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int DIL, bool MFMA>
int run(int B, int T, int reps)
{
    const int Ln = T * 8;
    const size_t nx = (size_t)B * 32 * Ln, nk = (size_t)B * T * fd::KREC;
    float *x, *skip, *out, *kp, *wref, *cb, *w16;
    int *flag;
    CK(hipMalloc(&w16, 12288)); CK(hipMalloc(&flag, 512)); CK(hipMemset(flag, 0, 512));
    {   // fp16 weight pieces of plausible magnitude
        std::vector<unsigned short> w(6144);
        unsigned sd = 777u;
        for (auto &v : w) { sd = sd * 1664525u + 1013904223u; v = (unsigned short)(((sd >> 16) & 0x8000u) | ((8u + ((sd >> 8) % 6u)) << 10) | ((sd >> 20) & 0x3FFu)); }
        CK(hipMemcpy(w16, w.data(), 12288, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&skip, nx * 4)); CK(hipMalloc(&out, nx * 4)); CK(hipMalloc(&kp, nk * 4));
    CK(hipMalloc(&wref, 3072 * 4)); CK(hipMalloc(&cb, 32 * 4));
    std::vector<float> h(std::max(nx, nk));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CK(hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(skip, h.data(), nx * 4, hipMemcpyHostToDevice));
    const float kscale = getenv("KSCALE") ? atof(getenv("KSCALE")) : 0.05f;
    for (size_t i = 0; i < nk; ++i) h[i] *= kscale;
    CK(hipMemcpy(kp, h.data(), nk * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wref, h.data(), 3072 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(cb, h.data(), 32 * 4, hipMemcpyHostToDevice));
    dim3 grid((Ln + 31) / 32, B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
        if constexpr (MFMA) hipLaunchKernelGGL((fdk_fast::k_lvc_h8m<DIL>), grid, dim3(256), 0, 0, x, skip, out, kp, 1, (const float4 *)w16, wref, cb, flag, T, (const int *)nullptr);
        else hipLaunchKernelGGL((fdk_fast::k_lvc_h8<DIL>), grid, dim3(256), 0, 0, x, skip, out, kp, 1, wref, cb, T, (const int *)nullptr, (const int *)nullptr);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = 4.0 * B * T * (96.0 * 8 + 6208);
    int hf = 0; CK(hipMemcpy(&hf, flag, 4, hipMemcpyDeviceToHost));
    printf("lvc_h8<%d> %s B=%d T=%d: %.1f us  %.0f GB/s (algorithmic)  flag=%d\n", DIL, MFMA ? "mfma" : "valu", B, T, us, bytes / us / 1e3, hf);
    hipFree(x); hipFree(skip); hipFree(out); hipFree(kp);
    return 0;
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 864;
    run<1, false>(B, T, 20);
    run<27, false>(B, T, 20);
    run<1, true>(B, T, 20);
    run<27, true>(B, T, 20);
    return 0;
}
