// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs? A[i][k] = 2^-20 (subnormal) for k==0, B[k][j] = 1024 → D = 2^-10 if honoured, 0 if flushed.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, unsigned short abits) {
    f16x8 a, b;
    _Float16 av = __builtin_bit_cast(_Float16, abits);
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    if ((threadIdx.x >> 4) == 0) { a[0] = av; b[0] = (_Float16)1024.f; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    // packed multiply of a subnormal-producing scale as well
    f16x8 s = a * (_Float16)(1.0f / 2048.f);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)s[0]; }
}
int main() {
    float *d, h[2];
    hipMalloc(&d, 8);
    unsigned short cases[] = {0x0010 /* 2^-20 subnormal */, 0x0001 /* 2^-24 */, 0x0400 /* 2^-14 min normal */, 0x2000 /* 2^-7: times 2^-11 -> 2^-18 subnormal */};
    for (unsigned short c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=0x%04x  mfma(a*1024)=%g  pk_mul(a/2048)=%g\n", c, h[0], h[1]);
    }
    return 0;
}
