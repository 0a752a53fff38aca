// poison.hip -- a second process that keeps leaving NaN bit patterns in every CU's LDS and vector registers for SECONDS seconds.
// Run next to a bit-reproducibility loop of the library (tools/flake_hunt.py): a kernel that reads LDS or a register it never wrote
// then sees NaNs instead of whatever its own previous launch left there, and the loop reports mismatches at once.
// Usage: tools/ubench/poison SECONDS [LDS_BYTES per poisoning workgroup = 65536] [workgroups per CU = 2]
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_poison_lds(unsigned *sink, unsigned pattern, int n)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < n; i += 256) lds[i] = pattern;
    __syncthreads();
    if (lds[(threadIdx.x * 7) % n] == 12345u) sink[0] = 1;      // keep the stores alive
}

#define P8(b) "v_mov_b32 v" #b "0, %0\n v_mov_b32 v" #b "1, %0\n v_mov_b32 v" #b "2, %0\n v_mov_b32 v" #b "3, %0\n v_mov_b32 v" #b "4, %0\n" \
              "v_mov_b32 v" #b "5, %0\n v_mov_b32 v" #b "6, %0\n v_mov_b32 v" #b "7, %0\n v_mov_b32 v" #b "8, %0\n v_mov_b32 v" #b "9, %0\n"
__global__ void __launch_bounds__(256, 2) k_poison_vgpr(unsigned *sink, unsigned pattern)
{
    // v10 .. v249 <- pattern (the kernel is declared to use 250 registers through the clobber list)
    asm volatile(P8(1) P8(2) P8(3) P8(4) P8(5) P8(6) P8(7) P8(8) P8(9) P8(10) P8(11) P8(12) P8(13) P8(14) P8(15) P8(16) P8(17) P8(18) P8(19)
                 P8(20) P8(21) P8(22) P8(23) P8(24)
                 :: "s"(pattern)
                 : "v10", "v19", "v20", "v29", "v50", "v99", "v100", "v150", "v199", "v200", "v249");
    if (pattern == 12345u) sink[0] = 1;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int lds_bytes = argc > 2 ? atoi(argv[2]) : 65536;      // LDS per poisoning workgroup
    const int wg_per_cu = argc > 3 ? atoi(argv[3]) : 2;
    unsigned *sink;
    CK(hipMalloc(&sink, 64));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    CK(hipFuncSetAttribute((const void *)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 50; ++k) {
            hipLaunchKernelGGL(k_poison_lds, dim3(prop.multiProcessorCount * wg_per_cu), dim3(256), lds_bytes, 0, sink, 0x7FC07FC0u, lds_bytes / 4);
            hipLaunchKernelGGL(k_poison_vgpr, dim3(prop.multiProcessorCount * 4), dim3(256), 0, 0, sink, 0x7FC07FC0u);
            launches += 2;
        }
        CK(hipDeviceSynchronize());
    }
    printf("poison: %ld launches in %.1f s\n", launches, seconds);
    return 0;
}
