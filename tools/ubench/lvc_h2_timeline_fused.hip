// lvc_h2_timeline_fused.hip -- phase timeline of the FUSED hop-256 variants of k_lvc_h2 (round 6): VARIANT 1 = the first layer of the block
// with the ConvTranspose inside (UP = 4), VARIANT 2 = the last layer with final_conv folded in (FINAL), VARIANT 0 = the plain layer with
// the same reduced set of stamps.  Per workgroup 10 x int64: slots 0 start, 1..4 the variant's own phases (FD_STAMP_X), 5 staging barrier,
// 6 LVC start, 7 end; HW_ID, XCC_ID.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-honor-nans -DFD_LVC_TIMELINE -DFD_TL_VARIANT=1 -o lvc_h2_timeline_up lvc_h2_timeline_fused.hip
#include "../../fastdiff_amd/csrc/fd_kernels_lvc.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
bool fd_prof_stamps(const fdk::Launch &, const char *, hipEvent_t *, hipEvent_t *) { return false; }
void fd_prof_begin(const fdk::Launch &, const char *) {}
void fd_prof_end(const fdk::Launch &) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const char *outp = argc > 1 ? argv[1] : "timeline.bin";
    const int B = argc > 2 ? atoi(argv[2]) : 8, T = argc > 3 ? atoi(argv[3]) : 864, HOP = 256, Ln = T * HOP;
    const size_t nx = (size_t)B * 32 * Ln, nk = (size_t)B * T * fd::KREC;
    float *x, *skip, *out, *kp, *wpack, *wref, *cb, *eps, *ff, *up16, *upb;
    int *flag;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&skip, nx * 4)); CK(hipMalloc(&out, nx * 4)); CK(hipMalloc(&kp, nk * 4));
    CK(hipMalloc(&wpack, 3072 * 4)); CK(hipMalloc(&wref, 3072 * 4)); CK(hipMalloc(&cb, 32 * 4)); CK(hipMalloc(&flag, 256)); CK(hipMemset(flag, 0, 256));
    CK(hipMalloc(&eps, (size_t)B * Ln * 4)); CK(hipMemset(eps, 0, (size_t)B * Ln * 4)); CK(hipMalloc(&ff, 64 * 16)); CK(hipMalloc(&up16, 8 * 2 * 4 * 64 * 16)); CK(hipMalloc(&upb, 32 * 4));
    std::vector<float> h(std::max(nx, nk));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CK(hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(skip, h.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ff, h.data(), 64 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(upb, h.data(), 32 * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nk; ++i) h[i] *= 0.05f;
    CK(hipMemcpy(kp, h.data(), nk * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wref, h.data(), 3072 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(cb, h.data(), 32 * 4, hipMemcpyHostToDevice));
    {
        std::vector<unsigned short> w16(8 * 2 * 4 * 64 * 8);
        unsigned s = 777u;
        for (auto &v : w16) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 16) & 0x8000u) | ((8u + ((s >> 8) % 6u)) << 10) | ((s >> 20) & 0x3FFu)); }
        CK(hipMemcpy(wpack, w16.data(), 6144 * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(up16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    }
    dim3 grid(((Ln + 255) / 256 + 7) / 8 * 8, B);
    const size_t nwg = (size_t)grid.x * grid.y;
    long long *tl;
    CK(hipMalloc(&tl, nwg * 80)); CK(hipMemset(tl, 0, nwg * 80));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(fdk_fast::fd_tl), &tl, sizeof(tl)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0, 0));
#if FD_TL_VARIANT == 1
        // xin = the block's INPUT [B][32][Ln / 4]
        hipLaunchKernelGGL((fdk_fast::k_lvc_h2<256, 1, false, 4>), grid, dim3(256), 0, 0, x, skip, out, kp, 0, (const float4 *)wpack, wref, cb, flag, T, (const int *)nullptr, (float *)nullptr, (const float4 *)nullptr, (const float4 *)up16, (const float *)upb, flag + 1);
#elif FD_TL_VARIANT == 2
        hipLaunchKernelGGL((fdk_fast::k_lvc_h2<256, 27, true>), grid, dim3(256), 0, 0, x, skip, out, kp, 3, (const float4 *)wpack, wref, cb, flag, T, (const int *)nullptr, eps, (const float4 *)ff, (const float4 *)nullptr, (const float *)nullptr, (int *)nullptr);
#else
        hipLaunchKernelGGL((fdk_fast::k_lvc_h2<256, 27, false>), grid, dim3(256), 0, 0, x, skip, out, kp, 1, (const float4 *)wpack, wref, cb, flag, T, (const int *)nullptr, (float *)nullptr, (const float4 *)nullptr, (const float4 *)nullptr, (const float *)nullptr, (int *)nullptr);
#endif
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d launch %d: %.1f us (with stamps)\n", FD_TL_VARIANT, rep, ms * 1e3);
    }
    std::vector<long long> d(nwg * 10);
    CK(hipMemcpy(d.data(), tl, nwg * 80, hipMemcpyDeviceToHost));
    FILE *f = fopen(outp, "wb");
    if (!f) { printf("cannot write %s\n", outp); return 1; }
    fwrite(d.data(), 8, d.size(), f);
    fclose(f);
    printf("wrote %zu workgroups (grid %u x %u) to %s\n", nwg, grid.x, grid.y, outp);
    return 0;
}
