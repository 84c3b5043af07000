// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  (needed by the f16x2 split: the low plane of a value
// below 2^-3 is an fp16 subnormal)   hipcc --offload-arch=gfx950 -O3 mfma_f16_denorm.hip -o mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__global__ void k(float* out, float av, float bv) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
    f32x2 x = {av * 3.3f, -av * 1.7f};
    f16x2 h = __builtin_convertvector(x, f16x2);
    if (threadIdx.x == 0) { out[1] = (float)h[0]; out[2] = (float)h[1]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float as[] = {1.0f, 0x1p-14f, 0x1p-15f, 0x1p-20f, 0x1p-24f};
    for (float av : as) {
        k<<<1, 64>>>(d, av, 1.0f);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a = %g (2^%d): D = %g, expected %g;  cvt_pk(3.3a, -1.7a) = %g %g\n", av, (int)std::log2(av), h[0], 16.0 * av, h[1], h[2]);
    }
    k<<<1, 64>>>(d, 0x1p-20f, 0x1p-10f);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("2^-20 x 2^-10: D = %g, expected %g\n", h[0], 16.0 * 0x1p-30);
    return 0;
}
