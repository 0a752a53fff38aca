// f16_probe.hip -- does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs, and does v_cvt_f16_f32 produce them?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float *out, float a_in, float b_in)
{
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)a_in; b[q] = (_Float16)b_in; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; union { _Float16 h; unsigned short u; } u; u.h = a[0]; out[2] = (float)u.u; }
}
int main()
{
    float *d, h[3];
    hipMalloc(&d, 12);
    const float as[4] = {9.5367431640625e-07f /*2^-20*/, 5.9604644775390625e-08f /*2^-24*/, 3.0517578125e-05f /*2^-15*/, 1.0f};
    for (int i = 0; i < 4; ++i) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, as[i], 1024.0f);
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g (fp16 bits 0x%04x, back %g) x 1024 x16 = %g   expected %g\n", as[i], (unsigned)h[2], h[1], h[0], as[i] * 1024.0 * 16);
    }
    return 0;
}
