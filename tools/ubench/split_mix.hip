// fp32 -> two fp16 planes (planes.hpp, NS = 2): the 6-op split (cvt_pk, 2 x cvt_f32_f16, 2 x sub, cvt_pk) against
// the v_fma_mix form (cvt_pk or 2 x v_fma_mixlo/hi_f16 with a power-of-two scale, 2 x v_fma_mix_f32 residuals that read
// the fp16 halves directly, cvt_pk): bitwise comparison on random and edge values, then VALU-only issue rate and the
// rate next to MFMAs.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize split_mix.hip -o split_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_std(float x0, float x1, unsigned& h, unsigned& l) {
    const f16x2 hh = __builtin_convertvector(f32x2{x0, x1}, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const float r0 = x0 - (float)hh[0], r1 = x1 - (float)hh[1];
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}
// residual x*s - half(h) in one op: the fp16 source is read straight from the packed plane word
__device__ __forceinline__ float resid_lo(float x, float s, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(h));
    return r;
}
__device__ __forceinline__ float resid_hi(float x, float s, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(h));
    return r;
}
__device__ __forceinline__ void split_mix(float x0, float x1, unsigned& h, unsigned& l) {      // unscaled: 4 ops
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, f16x2));
    const float r0 = resid_lo(x0, 1.0f, h), r1 = resid_hi(x1, 1.0f, h);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}
__device__ __forceinline__ void split_mix_scaled(float x0, float x1, float s, unsigned& h, unsigned& l) {   // 5 ops
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "v"(s));
    const float r0 = resid_lo(x0, s, h), r1 = resid_hi(x1, s, h);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}

__global__ void check(const float* x, int n, float s, unsigned* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned h, l, h2, l2, h3, l3, h4, l4;
    split_std(x[2 * i], x[2 * i + 1], h, l);
    split_mix(x[2 * i], x[2 * i + 1], h2, l2);
    split_std(x[2 * i] * s, x[2 * i + 1] * s, h3, l3);
    split_mix_scaled(x[2 * i], x[2 * i + 1], s, h4, l4);
    out[4 * i] = (h != h2) | ((l != l2) << 1);
    out[4 * i + 1] = (h3 != h4) | ((l3 != l4) << 1);
    out[4 * i + 2] = h4; out[4 * i + 3] = l4;
}

template <int MODE, int NMFMA>
__global__ void rate(float* out, int iters, float s) {
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 1.0f + threadIdx.x * 1e-3f + j;
    unsigned accu = 0;
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 xa = {threadIdx.x, 3u, 5u, 7u}, ya = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            unsigned h, l;
            if constexpr (MODE == 0) split_std(v[j], v[j + 1], h, l);
            else if constexpr (MODE == 1) split_mix(v[j], v[j + 1], h, l);
            else split_mix_scaled(v[j], v[j + 1], s, h, l);
            accu ^= h + l;
            v[j] = __uint_as_float((__float_as_uint(v[j]) & 0x3fffffffu) ^ (h >> 7));     // keep the chain data dependent, cheap
            if constexpr (NMFMA > 0) {
                if ((j & 3) == 0) {
#pragma unroll
                    for (int q = 0; q < NMFMA; ++q)
                        acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xa), __builtin_bit_cast(f16x8, ya), acc[q & 1], 0, 0, 0);
                }
            }
        }
    }
    float sum = 0.f;
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) sum += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum + (float)accu + v[0];
}

template <int MODE, int NMFMA>
void run(const char* name) {
    float* d;
    hipMalloc(&d, 256 * 512 * sizeof(float));
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 4000;
    rate<MODE, NMFMA><<<256, 512>>>(d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    rate<MODE, NMFMA><<<256, 512>>>(d, iters, 0.5f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    // per SIMD: 2 waves x iters x 8 pairs
    printf("%-28s %.3f ms  = %.2f ns per pair-split per wave (%.1f cycles @2.4 GHz, 2 waves/SIMD)\n", name, ms, ms * 1e6 / (iters * 8.0),
           ms * 1e6 / (iters * 8.0) * 2.4);
    hipFree(d);
}

int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int e = rand() % 60 - 40;
        h[i] = ldexpf((float)rand() / RAND_MAX * 2.f - 1.f, e);
    }
    const float edge[] = {0.f, -0.f, 65504.f, 65520.f, 1e5f, -1e5f, 6.1e-5f, 5.96e-8f, 1e-10f, 1.f, -1.f, 0.333333f, 32768.f, 2047.5f, 1e30f, 3.4e38f};
    for (int i = 0; i < 16; ++i) h[i] = edge[i];
    float* dx; unsigned* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 2 * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    for (float s : {1.0f, 0.0009765625f, 64.0f, 16384.0f}) {
        check<<<n / 2 / 256, 256>>>(dx, n, s, dout);
        std::vector<unsigned> o(n * 2);
        hipMemcpy(o.data(), dout, n * 2 * 4, hipMemcpyDeviceToHost);
        long bad_unscaled = 0, bad_scaled = 0;
        for (int i = 0; i < n / 2; ++i) { bad_unscaled += o[4 * i] != 0; bad_scaled += o[4 * i + 1] != 0; }
        printf("scale %g: mix split differs from the 6-op split on %ld pairs (unscaled), %ld pairs (scaled) of %d\n", s, bad_unscaled, bad_scaled, n / 2);
        if (bad_scaled || bad_unscaled) {
            int shown = 0;
            for (int i = 0; i < n / 2 && shown < 6; ++i)
                if (o[4 * i] | o[4 * i + 1]) { printf("   x = (%g, %g) flags %u %u  h %08x l %08x\n", h[2 * i], h[2 * i + 1], o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]); ++shown; }
        }
    }
    run<0, 0>("6-op split, VALU only");
    run<1, 0>("mix split (4 ops), VALU only");
    run<2, 0>("mix scaled (5 ops), VALU only");
    run<0, 2>("6-op split + 4 MFMA / pair-quad");
    run<1, 2>("mix split + MFMA");
    run<2, 2>("mix scaled + MFMA");
    return 0;
}
