// Operand planes of the matrix-pipe kernels (sa_mlp_bf16.hip, gemm_bf16.hip): every fp32 operand is represented by NS
// 16-bit planes and a product x*w by a few plane products accumulated in fp32 by v_mfma_f32_32x32x16_{bf16,f16}.
//   NS = 1  "bf16"    one bf16 plane (round to nearest even), 1 product                      -- BASELINE.json config 3
//   NS = 2  "f16x2"   x = h + l, two fp16 planes (h = rne(x), l = rne(x - h)): 11 + 11 mantissa bits; products
//                     hl, lh, hh (the dropped ll term is O(2^-22)); |x| must stay below 65504, values whose low
//                     plane is an fp16 subnormal keep an absolute error <= 2^-25 (the MFMA honours subnormals:
//                     tools/ubench/mfma_f16_denorm.hip)
//   NS = 3  "bf16x3"  x = h + m + l, three bf16 planes by exact truncation (8 + 8 + 8 bits); products hl, lh, mm,
//                     hm, mh, hh (dropped terms O(2^-24)); full fp32 range
//   NS = 4  "f16"     [r6] ONE fp16 plane (h = rne(x), 11 mantissa bits: 8 x finer than bf16), 1 product, WITH the range
//                     machinery of f16x2 (per-window power-of-two scaling from the range records, W / u planes) -- the
//                     reduced-precision mode that survives a trained checkpoint (BASELINE.json config 3).  NS = 4 is a mode
//                     CODE, not a plane count: plane_count(4) = 1, and the tile images have the bf16 mode's geometry.
// Smallest terms are accumulated first.  The same splits are applied to the weights on the host (csrc/pack.hip).
//
// Range of f16x2 (fp16 has 5 exponent bits).  Weights: the host takes the planes of W / u, u a power of two (csrc/pack.hip:
// plane_unscale).  Activations: every tensor a contraction reads carries a per-window absolute maximum ("amax", the bit
// pattern of a non-negative float, kept up to date with integer atomicMax by the kernel that produces the tensor); the
// consumer multiplies its rows by the power of two that puts that maximum in [2^14, 2^15) before the split and multiplies the
// accumulated products back (both exact).  Values then never reach the fp16 overflow threshold, whatever the checkpoint or
// the input, and a value keeps its full 22 bits down to 2^-17 of the window's maximum (below that the absolute error is
// <= 2^-39 of the maximum).  The scale depends on the window only, so results do not depend on how a batch is sharded.
#pragma once
#include "common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// mode code -> number of 16-bit planes per operand / fp16 planes with range scaling (f16x2, f16)
constexpr int plane_count(int ns) { return ns == 4 ? 1 : ns; }
constexpr bool planes_f16(int ns) { return ns == 2 || ns == 4; }

template <int NS> struct Planes;
template <> struct Planes<1> { static constexpr int NPROD = 1; static constexpr int A[1] = {0}, B[1] = {0}; };
template <> struct Planes<2> { static constexpr int NPROD = 3; static constexpr int A[3] = {0, 1, 0}, B[3] = {1, 0, 0}; };
template <> struct Planes<4> { static constexpr int NPROD = 1; static constexpr int A[1] = {0}, B[1] = {0}; };
template <> struct Planes<3> { static constexpr int NPROD = 6; static constexpr int A[6] = {0, 2, 1, 0, 1, 0}, B[6] = {2, 0, 1, 1, 0, 0}; };

template <int NS>
__device__ __forceinline__ f32x16 mfma_planes(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (planes_f16(NS)) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the 16 x 16 x 32 form: D[16 x 16] += A[16 x 32] * B[32 x 16]; lane l holds A[i = l & 15][k = 8 (l >> 4) ..+8], B[k = 8 (l >> 4) ..+8][j = l & 15],
// D[i = 4 (l >> 4) + r][j = l & 15], r = 0..3.  Same FLOP per issue cycle as the 32 x 32 x 16 form with a quarter of the accumulator
// registers -- and 17-18 % more sustained throughput at the power limit (profiles/r6_mfma_ceiling.txt).
template <int NS>
__device__ __forceinline__ f32x4 mfma16_planes(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (planes_f16(NS)) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float relu_bits(float x) {     // max(x, 0) as one integer max: no canonicalisation op, -0 -> +0
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}
__device__ __forceinline__ unsigned pack_hi16(float x1, float x0) {     // upper halves of x1 : x0, one v_perm_b32
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float trunc_hi16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// two fp32 values -> NS packed plane words (element 0 in the low half-word).  Residual subtracts stay scalar: packed
// fp32 VALU ops are slower than two plain ones on gfx950 (the build also passes -fno-slp-vectorize).
template <int NS>
__device__ __forceinline__ void split_planes(float x0, float x1, unsigned (&o)[plane_count(NS)]) {
    if constexpr (NS == 1) {
        o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
    } else if constexpr (NS == 4) {
        o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, f16x2));      // v_cvt_pk_f16_f32 (RNE)
    } else if constexpr (NS == 2) {
        // 4 VALU ops per pair: v_cvt_pk_f16_f32, two v_fma_mix_f32 that read the fp16 halves of the packed high plane
        // directly (residual x - float(h), exact), v_cvt_pk_f16_f32 -- bit-identical to the cvt / cvt-back / subtract form
        // (6 ops) and 9-17 % faster (tools/ubench/split_mix.hip)
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, f16x2));   // RNE
        o[0] = h;
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(h));
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(h));
        o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
    } else {
        o[0] = pack_hi16(x1, x0);
        const float r0 = x0 - trunc_hi16(x0), r1 = x1 - trunc_hi16(x1);                           // exact
        o[1] = pack_hi16(r1, r0);
        const float q0 = r0 - trunc_hi16(r0), q1 = r1 - trunc_hi16(r1);
        o[2] = pack_hi16(q1, q0);
    }
}

// ReLU of two packed bf16 values in one v_pk_max_i16: a negative float's bf16 pattern is a negative int16 (-0 included)
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_pk_bf16(unsigned v) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), s16x2{0, 0}));
}

// packed fp16, NON-NEGATIVE values (after relu_pk_bf16): clamp +inf / NaN patterns (> 0x7bff as int16) to the largest finite fp16
__device__ __forceinline__ unsigned sat_pk_f16(unsigned v) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(s16x2, v), s16x2{0x7bff, 0x7bff}));
}
// NOTE on inline asm in these kernels: an asm block must never be the FIRST consumer of an MFMA result -- the wait states between an
// MFMA's register write and a VALU read are inserted by the compiler, which does not see into asm (a v_fma_mixlo_f16 asm on raw
// accumulators read stale values, round 6).  split_planes<2>'s asm is fed by ordinary VALU results (scaled / clamped values) only.

// ---- f16x2 activation range (see the header comment) --------------------------------------------------------------------
// power of two s such that a * s lies in [2^14, 2^15) for the non-negative float with bit pattern `amax_bits`
// (clamped to 2^73 for tiny / zero maxima; 1 for inf / NaN: garbage in, garbage out, as in the reference)
__device__ __forceinline__ float f16x2_scale(unsigned amax_bits) {
    const int E = (int)((amax_bits >> 23) & 0xffu);
    const int sb = (E == 255) ? 127 : min(268 - E, 200);
    return __uint_as_float((unsigned)sb << 23);
}
__device__ __forceinline__ float pow2_inverse(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }
// max(|v|) bookkeeping: non-negative floats order like their bit patterns
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
// ReLU that also saturates at the largest fp16 (one v_med3_f32, the cost of the integer-max ReLU): a bound the caller broke
// (neighbours farther than the contract's dmax) clamps instead of producing inf planes
__device__ __forceinline__ float relu_sat_f16(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 65504.f); }
