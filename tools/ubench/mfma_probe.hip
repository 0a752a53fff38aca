// mfma_probe.hip -- which ingredient of the GEMM inner loop costs MFMA issue rate on gfx950?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

constexpr int LD = 260;

// MODE 0: 1 chain, operands in registers      1: 1 chain, A from LDS     2: 2 chains, A from LDS
// MODE 3: 1 chain, A from LDS + 16 stores per 96 MFMA   4: 4 chains regs   5: 2 chains LDS + stores
template <int MODE>
__global__ void __launch_bounds__(256, 2) probe(const float *__restrict__ w, float *__restrict__ out, int iters)
{
    __shared__ float hs[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    for (int i = tid; i < 64 * LD; i += 256) hs[i] = (float)(i % 7) * 0.01f;
    float wb[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) wb[i] = w[i * 64 + lane];
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float *hw = hs + hi * LD + l31;
    float *o = out + (size_t)blockIdx.x * 256 * 16 * 8 + tid;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const float *ht = hw + (it & 7) * 32;
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 96; ++s) acc[0] = mfma32(wb[(s + 1) % 96], wb[s], acc[0]);
        } else if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int s = 0; s < 96; ++s) acc[0] = mfma32(ht[((2 * s) & 63) * LD + (s >> 5)], wb[s], acc[0]);
        } else if (MODE == 2 || MODE == 5) {
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                acc[0] = mfma32(ht[((2 * s) & 63) * LD + (s >> 5)], wb[s], acc[0]);
                acc[1] = mfma32(ht[((2 * s) & 63) * LD + (s >> 5) + 32], wb[s], acc[1]);
            }
        } else if (MODE == 4) {
#pragma unroll
            for (int s = 0; s < 24; ++s) {
                acc[0] = mfma32(wb[(s + 1) % 96], wb[s], acc[0]);
                acc[1] = mfma32(wb[(s + 2) % 96], wb[s], acc[1]);
                acc[2] = mfma32(wb[(s + 3) % 96], wb[s], acc[2]);
                acc[3] = mfma32(wb[(s + 4) % 96], wb[s], acc[3]);
            }
        }
        if (MODE == 6) {        // conv-like: 48 steps, weights in regs as A, B = lrelu(LDS)
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                float v = hs[(((2 * s) & 31) + hi) * LD + 4 + l31 + (it & 7) * 16 + ((s >> 4) - 1) * 9];
                v = __builtin_amdgcn_fmed3f(v, v * 0.2f, __builtin_inff());
                acc[0] = mfma32(wb[s], v, acc[0]);
            }
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                float v = hs[(((2 * s) & 31) + hi) * LD + 36 + l31 + (it & 7) * 16 + ((s >> 4) - 1) * 9];
                v = __builtin_amdgcn_fmed3f(v, v * 0.2f, __builtin_inff());
                acc[1] = mfma32(wb[s], v, acc[1]);
            }
        }
        if (MODE == 7) {        // lvc-like: 48 steps x 2 row tiles sharing B from LDS, A in regs
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                const float v = hs[(((2 * s) & 31) + hi) * LD + 4 + l31 + (it & 7) * 16 + (s >> 4)];
                acc[0] = mfma32(wb[s], v, acc[0]);
                acc[1] = mfma32(wb[48 + s], v, acc[1]);
            }
        }
        if (MODE == 3 || MODE == 5) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)((it & 7) * 16 + r) * 256] = acc[0][r] + acc[1][r];
        }
    }
    float sacc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[j][r];
    if (sacc == 123.456f) out[tid] = sacc;
}

template <int MODE>
int run(const char *name, const float *w, float *out, int grid)
{
    const int iters = 400;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, w, out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, w, out, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * 4 * iters * 96 * 4096.0;
    printf("%-44s grid=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, grid, ms, flops / ms / 1e9);
    return 0;
}

int main()
{
    float *w, *out;
    CK(hipMalloc(&w, 96 * 64 * 4));
    CK(hipMalloc(&out, (size_t)4096 * 256 * 16 * 8 * 4));
    std::vector<float> hw(96 * 64, 0.001f);
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    for (int grid : {512, 2048}) {
        run<0>("0: 1 chain, regs", w, out, grid);
        run<4>("4: 4 chains, regs", w, out, grid);
        run<1>("1: 1 chain, A from LDS", w, out, grid);
        run<2>("2: 2 chains, A from LDS", w, out, grid);
        run<3>("3: 1 chain, LDS + 16 stores/96", w, out, grid);
        run<5>("5: 2 chains, LDS + 16 stores/96", w, out, grid);
        run<6>("6: conv-like (lrelu(LDS) as B), 2x48", w, out, grid);
        run<7>("7: lvc-like (2 row tiles share B)", w, out, grid);
    }
    return 0;
}
