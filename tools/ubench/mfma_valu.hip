// Does VALU work overlap MFMA work on one SIMD?  (a) two waves per SIMD, one issuing MFMAs and one issuing VALU ops;
// (b) one instruction stream with KV VALU ops after every MFMA.   hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, x, y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
#define VALU(v, c) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v) : "v"(c))

// roles: bit0 of mask -> waves 0-3 run the MFMA loop, bit1 -> waves 4-7 run the VALU loop
__global__ __launch_bounds__(512) void roles(long long* tm, float* out, int it_m, int it_v, int mask) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 x = {threadIdx.x, 3u, 5u, 7u}, y = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    float v[4] = {1.f + threadIdx.x, 2.f, 3.f, 4.f}, c = 1e-7f * threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mask & 1)
            for (int i = 0; i < it_m; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) MFMA(acc, x, y);
            }
    } else {
        if (mask & 2)
            for (int i = 0; i < it_v; ++i) {
#pragma unroll
                for (int u = 0; u < 64; ++u) VALU(v[u & 3], c);
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = v[0] + v[1] + v[2] + v[3];
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) tm[wave] = t1 - t0;
}

template <int KV>
__global__ __launch_bounds__(512) void same(long long* tm, float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 x = {threadIdx.x, 3u, 5u, 7u}, y = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    float v[4] = {1.f + threadIdx.x, 2.f, 3.f, 4.f}, c = 1e-7f * threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            MFMA(acc, x, y);
#pragma unroll
            for (int k = 0; k < KV; ++k) VALU(v[k & 3], c);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = v[0] + v[1] + v[2] + v[3];
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) tm[wave] = t1 - t0;
}

long long* tm; float* out;
void report(const char* name, double n_mfma, double n_valu) {
    long long h[8];
    hipDeviceSynchronize();
    hipMemcpy(h, tm, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s ticks: w0 %8lld  w4 %8lld", name, h[0], h[4]);
    if (n_mfma > 0) printf("   %.1f ticks/MFMA(w0)", h[0] / n_mfma);
    if (n_valu > 0) printf("   %.2f ticks/VALU(w4)", h[4] / n_valu);
    printf("\n");
}
template <int KV>
void run_same(int threads) {
    const int iters = 1000;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    same<KV><<<256, threads>>>(tm, out, 10);
    hipEventRecord(s);
    same<KV><<<256, threads>>>(tm, out, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    long long h[8]; hipMemcpy(h, tm, sizeof(h), hipMemcpyDeviceToHost);
    printf("same-stream KV=%2d waves/SIMD=%d : %.1f ticks per (MFMA+KV VALU) per wave, %.2f ns per MFMA per SIMD\n", KV, threads / 256,
           h[0] / (iters * 8.0), ms * 1e6 / (iters * 8.0 * (threads / 256)));
}
int main() {
    hipMalloc(&tm, 64); hipMalloc(&out, 256 * 512 * 4);
    hipMemset(tm, 0, 64);
    const int it_m = 1000, it_v = 1000;     // 8000 MFMAs vs 64000 VALU ops
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int mask = 1; mask <= 3; ++mask) {
        roles<<<256, 512>>>(tm, out, 10, 10, mask);
        hipEventRecord(s);
        roles<<<256, 512>>>(tm, out, it_m, it_v, mask);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        char name[64]; snprintf(name, 64, "roles mask=%d (%.3f ms)", mask, ms);
        report(name, (mask & 1) ? it_m * 8.0 : 0, (mask & 2) ? it_v * 64.0 : 0);
    }
    run_same<0>(256); run_same<0>(512);
    run_same<2>(256); run_same<2>(512);
    run_same<4>(256); run_same<4>(512);
    run_same<6>(256); run_same<6>(512);
    run_same<8>(256); run_same<8>(512);
    run_same<12>(512);
    return 0;
}
