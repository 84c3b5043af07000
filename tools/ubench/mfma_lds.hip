// How fast can v_mfma_f32_32x32x16_bf16 run when every group of 6 MFMAs needs 3 fresh ds_read_b128 fragments
// (the bf16x3 inner loop shape), 8 waves per CU (2 per SIMD)?   hipcc --offload-arch=gfx950 -O3 mfma_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int READS, int NACC, bool BARRIER>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 40960 / 4; i += 512) ((unsigned*)smem)[i] = 0x3f803f80u + i;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 h[3] = {{1u + lane, 2u, 3u, 4u}, {5u, 6u + lane, 7u, 8u}, {9u, 10u, 11u + lane, 12u}};
    const char* base = smem + (lane & 31) * 1264 + (lane >> 5) * 16;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 13; ++g) {
            u32x4 w[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) w[s] = (s < READS) ? *reinterpret_cast<const u32x4*>(base + s * 416 + g * 32) : h[s];
            f32x16& c = acc[g % NACC];
            c = mf(h[0], w[2], c); c = mf(h[2], w[0], c); c = mf(h[1], w[1], c);
            c = mf(h[0], w[1], c); c = mf(h[1], w[0], c); c = mf(h[0], w[0], c);
        }
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int READS, int NACC, bool BARRIER>
void run(const char* name) {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    const int iters = 400;
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    k<READS, NACC, BARRIER><<<256, 512, 40960>>>(d, 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    k<READS, NACC, BARRIER><<<256, 512, 40960>>>(d, iters);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    const double n_per_simd = (double)iters * 13 * 6 * 2;
    printf("%-34s: %.2f ns per MFMA per SIMD  -> %.0f%% of the 15.7 ns MFMA-only rate\n", name, ms * 1e6 / n_per_simd, 15.7 / (ms * 1e6 / n_per_simd) * 100);
    (void)hipFree(d);
}

int main() {
    run<0, 1, false>("no LDS reads, 1 acc");
    run<3, 1, false>("3 reads / 6 MFMA, 1 acc");
    run<3, 2, false>("3 reads / 6 MFMA, 2 acc");
    run<3, 1, true>("3 reads / 6 MFMA, 1 acc, barrier/78");
    run<1, 1, false>("1 read / 6 MFMA, 1 acc");
    return 0;
}
