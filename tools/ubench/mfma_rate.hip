// MFMA micro-benchmark: issue interval vs dependent-accumulator latency of v_mfma_f32_32x32x16_bf16 and
// v_mfma_f32_32x32x2_f32, with 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool BF16>
__global__ void k(float* out, int iters, unsigned seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 x = {seed + threadIdx.x, seed * 3u, seed * 5u, seed * 7u};
    u32x4 y = {seed ^ 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                if constexpr (BF16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool BF16>
void run(const char* name, int waves_per_simd) {
    float* d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    const int iters = 2000, threads = 256 * waves_per_simd, blocks = 256;   // one block per CU
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<NACC, BF16><<<blocks, threads>>>(d, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<NACC, BF16><<<blocks, threads>>>(d, iters, 1);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const double n_per_simd = (double)iters * 8 * NACC * waves_per_simd;
    const double ns_per = ms * 1e6 / n_per_simd;
    const double flop = (BF16 ? 32768.0 : 4096.0) * n_per_simd * 1024;
    printf("%-10s acc=%d waves/SIMD=%d : %.2f ns per MFMA per SIMD (%.1f cycles @2.4GHz)  %.0f TFLOP/s\n", name, NACC, waves_per_simd,
           ns_per, ns_per * 2.4, flop / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main() {
    run<1, true>("bf16 dep", 1);
    run<1, true>("bf16 dep", 2);
    run<2, true>("bf16 2acc", 1);
    run<4, true>("bf16 4acc", 1);
    run<4, true>("bf16 4acc", 2);
    run<1, false>("f32 dep", 1);
    run<1, false>("f32 dep", 2);
    run<4, false>("f32 4acc", 1);
    return 0;
}
