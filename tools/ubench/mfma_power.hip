// Sustained rate of v_mfma_f32_32x32x16_f16 on RANDOM vs ALL-ZERO operands, with the observed shader clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o gpurun_out/mfma_power && gpurun_out/mfma_power
// Why: DESIGN.md 3.2 claims that the matrix-pipe kernels of this library run at the rate the chip SUSTAINS on random data
// (~1.2 PFLOP/s of fp16 products, ~48 % of the 2.5 PFLOP/s datasheet peak) because the clock is set by power: the same
// instruction stream on zeros runs faster.  This program makes that checkable: a pure MFMA loop (no memory, no VALU in the
// loop, 4 independent accumulators, 1 or 2 waves per SIMD, every CU busy) timed with HIP events, and the shader clock measured
// as delta(s_memtime) / delta(s_memrealtime) x 100 MHz (s_memrealtime counts the constant 100 MHz reference clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Clk { unsigned long long core, ref; };

// PATTERN 3: v_mfma_f32_32x32x16_bf16 on the same bit patterns; 4: v_mfma_f32_32x32x2_f32 (the exact-fp32 mode's instruction).
// PATTERN 5 [r6]: v_mfma_f32_16x16x32_f16 -- the same FLOP per instruction-issue-cycle (16 x 16 x 32 x 2 = 16 384 FLOP in 4 passes against
//            32 768 in 8) with a QUARTER of the accumulator registers (4 instead of 16 per tile): does the smaller tile hold a higher
//            clock under the power limit?    Four independent 16 x 16 accumulators, operands as in pattern 0.
// PATTERN 0: four consecutive MFMAs share the B operand and take different A operands (what the loop nests of the library do: one
//            activation fragment against several weight fragments); 1: both operands change at every MFMA; 2: both operands fixed.
template <int WAVES, int PATTERN>
__global__ __launch_bounds__(256 * WAVES) void k(const f16x8* __restrict__ ops, float* __restrict__ out, Clk* __restrict__ clk, int iters) {
    f32x16 acc[4];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc4[4] = {};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 A[4], B[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A[i] = ops[(i * 2 + 0) * 64 + (threadIdx.x & 63)];
        B[i] = ops[(i * 2 + 1) * 64 + (threadIdx.x & 63)];
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a)
            {
                if constexpr (PATTERN == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(u + a) & 3], B[u], acc[a], 0, 0, 0);
                else if constexpr (PATTERN == 1) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(u + a) & 3], B[(u + 3 * a) & 3], acc[a], 0, 0, 0);
                else if constexpr (PATTERN == 2) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[0], acc[a], 0, 0, 0);
                else if constexpr (PATTERN == 5) {      // two 16 x 16 x 32 MFMAs = the FLOP of one 32 x 32 x 16
                    acc4[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[(u + a) & 3], B[u], acc4[a], 0, 0, 0);
                    acc4[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[(u + a + 1) & 3], B[(u + 1) & 3], acc4[a], 0, 0, 0);
                }
                else if constexpr (PATTERN == 3) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[(u + a) & 3]), __builtin_bit_cast(bf16x8, B[u]), acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, (unsigned)(0x3f000000u | (__builtin_bit_cast(unsigned short, A[(u + a) & 3][0]) << 7))),
                                                                   __builtin_bit_cast(float, (unsigned)(0x3f000000u | (__builtin_bit_cast(unsigned short, B[u][1]) << 7))), acc[a], 0, 0, 0);
            }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
        s += acc4[a][0] + acc4[a][1] + acc4[a][2] + acc4[a][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x].core = c1 - c0; clk[blockIdx.x].ref = r1 - r0; }
}

template <int WAVES, int PATTERN = 0>
static void run(const char* label, const f16x8* d_ops, int ncu) {
    float* d_out; Clk* d_clk;
    hipMalloc(&d_out, (size_t)ncu * 256 * WAVES * sizeof(float));
    hipMalloc(&d_clk, ncu * sizeof(Clk));
    const int iters = 60000;                                      // ~50-100 ms: long enough for the power management to settle
    k<WAVES, PATTERN><<<ncu, 256 * WAVES>>>(d_ops, d_out, d_clk, 2000);
    hipDeviceSynchronize();
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    float best = 1e30f, worst = 0.f; double clk_mhz = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(s);
        k<WAVES, PATTERN><<<ncu, 256 * WAVES>>>(d_ops, d_out, d_clk, iters);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        best = ms < best ? ms : best; worst = ms > worst ? ms : worst;
        std::vector<Clk> h(ncu);
        hipMemcpy(h.data(), d_clk, ncu * sizeof(Clk), hipMemcpyDeviceToHost);
        double acc = 0; for (auto& c : h) acc += (double)c.core / (double)c.ref * 100.0;
        clk_mhz = acc / ncu;                                      // last repetition: the settled clock
    }
    const double mfma = (double)iters * 16 * 4 * WAVES * ncu;     // per SIMD x 4 SIMDs x CUs
    const double flop = mfma * (PATTERN == 4 ? 4096.0 : 32768.0);
    const double peak = PATTERN == 4 ? 157.3e12 : 2.5e15;
    printf("%-8s waves/SIMD=%d : %8.3f ms (worst of 5: %8.3f)  %7.1f TFLOP/s  = %.3f of %.1f dense peak   shader clock %.0f MHz  "
           "(%.2f cycles per MFMA per SIMD)\n", label, WAVES, best, worst, flop / (best * 1e-3) / 1e12, flop / (best * 1e-3) / peak, peak / 1e12, clk_mhz,
           best * 1e-3 * clk_mhz * 1e6 / ((double)iters * 16 * WAVES));
    hipFree(d_out); hipFree(d_clk);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, nominal clock %d MHz\n", prop.gcnArchName, ncu, prop.clockRate / 1000);
    std::vector<_Float16> h(8 * 64 * 8);
    f16x8* d_ops; hipMalloc(&d_ops, h.size() * sizeof(_Float16));
    for (int pass = 0; pass < 2; ++pass) {
        srand(12345);
        for (auto& v : h) {
            const float r = (float)rand() / RAND_MAX * 2.f - 1.f;  // uniform (-1, 1): full-width fp16 mantissas, zero mean
            v = (_Float16)(pass == 0 ? 0.f : r);      // (pass 0: "zero f32" still has the constant exponent bits: 0.5 x 0.5)
        }
        hipMemcpy(d_ops, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        run<1>(pass == 0 ? "zeros" : "random", d_ops, ncu);
        run<2>(pass == 0 ? "zeros" : "random", d_ops, ncu);
        if (pass == 1) {
            run<2, 1>("rnd A+B", d_ops, ncu);          // both operands change at every MFMA
            run<2, 2>("rnd fix", d_ops, ncu);          // the same random operands at every MFMA
            run<2, 3>("rnd bf16", d_ops, ncu);         // v_mfma_f32_32x32x16_bf16, the same bit patterns read as bf16
            run<2, 4>("rnd f32", d_ops, ncu);          // v_mfma_f32_32x32x2_f32, random mantissas in [0.5, 1)
            run<2, 5>("rnd 16x16", d_ops, ncu);        // v_mfma_f32_16x16x32_f16 (two per 32 x 32 x 16's worth of FLOP), random operands
            run<1, 5>("rnd 16x16", d_ops, ncu);
        } else {
            run<2, 4>("zero f32", d_ops, ncu);
            run<2, 5>("zero 16x16", d_ops, ncu);
        }
    }
    return 0;
}
