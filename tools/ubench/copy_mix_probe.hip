// copy_mix_probe.hip -- what the memory system gives hand-written kernels with the LVC layer's traffic mix (LABBOOK.md 3.1):
//   copy      out = a                                  1 read : 1 write, float4 per lane, linear
//   mix21     out = a + b                              2 reads : 1 write (x, skip -> x'), linear
//   lvc256    the global-memory instructions of k_lvc_h2<256,*> and nothing else: per workgroup one 256-column tile of 32
//             channels of x and skip (float4 per lane, 8 rows per wave), the frame's 24.8 KB predicted-kernel record share
//             (12 float4 per lane), and the dword stores of the MFMA result layout (32 lanes = one 128 B line per instruction)
//   lvc64     the same for hop 64 (one frame per wave: 24 float4 of record per lane)
// These are the ceilings the LVC kernels are priced against instead of torch.Tensor.copy_ (tools/bw_probe.py).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_copy(const float4 *__restrict__ a, float4 *__restrict__ out, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = a[i];
}
__global__ void __launch_bounds__(256) k_copy_x4(const float4 *__restrict__ a, float4 *__restrict__ out, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i + k * 256 < n4) ? a[i + k * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < n4) out[i + k * 256] = v[k];
}
__global__ void __launch_bounds__(256) k_mix21(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ out, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    float4 va[4], vb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        va[k] = (i + k * 256 < n4) ? a[i + k * 256] : make_float4(0, 0, 0, 0);
        vb[k] = (i + k * 256 < n4) ? b[i + k * 256] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i + k * 256 < n4) out[i + k * 256] = make_float4(va[k].x + vb[k].x, va[k].y + vb[k].y, va[k].z + vb[k].z, va[k].w + vb[k].w);
}

constexpr int KREC = 24832, KLAYER = 6144;

// HOP 256: workgroup = 256 columns = one frame; wave = (row tile mt, column half).  HOP 64: wave = one frame, both row tiles.
template <int HOP>
__global__ void __launch_bounds__(256, 2) k_lvc_traffic(const float *__restrict__ x, const float *__restrict__ skip, float *__restrict__ out,
                                                        const float *__restrict__ kpack, int layer, int T)
{
    constexpr int LT = HOP == 256 ? 1 : 2, LN = HOP == 256 ? 4 : 2;
    const int Ln = T * HOP, b = blockIdx.y, w0 = blockIdx.x * 256;
    if (w0 >= Ln) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int mt0 = HOP == 256 ? (wave & 1) : 0, lcw = HOP == 256 ? 128 * (wave >> 1) : 64 * wave;
    const int f = (w0 + lcw) / HOP;
    const float *rec = kpack + ((size_t)b * T + f) * KREC;
    const float4 *kp4 = reinterpret_cast<const float4 *>(rec + layer * KLAYER) + 2 * lane;
    float4 ka[LT][12];
#pragma unroll
    for (int m = 0; m < LT; ++m)
#pragma unroll
        for (int i = 0; i < 12; ++i) ka[m][i] = kp4[((mt0 + m) * 6 + (i >> 1)) * 128 + (i & 1)];
    const float *xr = x + ((size_t)b * 32 + wave * 8) * Ln + w0 + 4 * lane, *sr = skip + ((size_t)b * 32 + wave * 8) * Ln + w0 + 4 * lane;
    float4 xa[8], sa[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        xa[c] = *reinterpret_cast<const float4 *>(xr + (size_t)c * Ln);
        sa[c] = *reinterpret_cast<const float4 *>(sr + (size_t)c * Ln);
    }
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += xa[c].x + sa[c].x + xa[c].y + sa[c].y + xa[c].z + sa[c].z + xa[c].w + sa[c].w;
#pragma unroll
    for (int m = 0; m < LT; ++m)
#pragma unroll
        for (int i = 0; i < 12; ++i) acc += ka[m][i].x + ka[m][i].y + ka[m][i].z + ka[m][i].w;
    float *xo = out + (size_t)b * 32 * Ln + (size_t)(4 * hi) * Ln + w0 + lcw + l31;
#pragma unroll
    for (int nt = 0; nt < LN; ++nt)
#pragma unroll
        for (int m = 0; m < LT; ++m)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int chl = 16 * (mt0 + m) + (r & 3) + 8 * (r >> 2);
                xo[(size_t)chl * Ln + nt * 32] = acc + (float)r;
            }
}

template <typename F>
static double time_us(F launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3 / reps;
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 864;
    const size_t nx = (size_t)B * 32 * T * 256, nk = (size_t)B * T * KREC;
    float *x, *skip, *out, *kp;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&skip, nx * 4)); CK(hipMalloc(&out, nx * 4)); CK(hipMalloc(&kp, nk * 4));
    CK(hipMemset(x, 0, nx * 4)); CK(hipMemset(skip, 0, nx * 4)); CK(hipMemset(kp, 0, nk * 4));
    const size_t n4 = nx / 4;
    const double mb = nx * 4 / 1e6;
    double us;
    us = time_us([&] { hipLaunchKernelGGL(k_copy, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float4 *)x, (float4 *)out, n4); }, 20);
    printf("copy     (1 float4/lane)  %7.1f MB each way: %7.1f us  %.2f TB/s (r+w)\n", mb, us, 2 * nx * 4 / us / 1e6);
    us = time_us([&] { hipLaunchKernelGGL(k_copy_x4, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, (const float4 *)x, (float4 *)out, n4); }, 20);
    printf("copy     (4 float4/lane)  %7.1f MB each way: %7.1f us  %.2f TB/s (r+w)\n", mb, us, 2 * nx * 4 / us / 1e6);
    us = time_us([&] { hipLaunchKernelGGL(k_mix21, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, (const float4 *)x, (const float4 *)skip, (float4 *)out, n4); }, 20);
    printf("mix 2:1  (x + skip -> out) %6.1f MB x3:       %7.1f us  %.2f TB/s (2r+w)\n", mb, us, 3 * nx * 4 / us / 1e6);
    {
        const int Ln = T * 256;
        const double bytes = 4.0 * B * T * (96.0 * 256 + 6208);
        us = time_us([&] { hipLaunchKernelGGL(k_lvc_traffic<256>, dim3((Ln + 255) / 256, B), dim3(256), 0, 0, x, skip, out, kp, 1, T); }, 20);
        printf("lvc256 traffic only (x, skip, record share, MFMA-layout stores) %.1f MB: %7.1f us  %.2f TB/s = %.1f %% of 8 TB/s\n", bytes / 1e6, us, bytes / us / 1e6, bytes / us / 1e6 / 8 * 100);
    }
    {
        const int Ln = T * 64;
        const double bytes = 4.0 * B * T * (96.0 * 64 + 6208);
        us = time_us([&] { hipLaunchKernelGGL(k_lvc_traffic<64>, dim3((Ln + 255) / 256, B), dim3(256), 0, 0, x, skip, out, kp, 1, T); }, 20);
        printf("lvc64  traffic only                                              %.1f MB: %7.1f us  %.2f TB/s = %.1f %% of 8 TB/s\n", bytes / 1e6, us, bytes / us / 1e6, bytes / us / 1e6 / 8 * 100);
    }
    hipFree(x); hipFree(skip); hipFree(out); hipFree(kp);
    return 0;
}
