// AttentionBlock of the Ev2Hands segmentation head for gfx950 (reference: model/TEHNet.py:13-27):
//   sim = softmax_over_classes( 256^-0.5 * key[B,4,N] @ query^T[B,N,256] )   -> [B,4,256]
//   ctx = sim @ value[B,256,N]                                                -> [B,4,N]
// Point-major operands: key = logits_pm [B*N][4], query/value rows of 256 channels.  Both are
// HBM-streaming reductions (1 KB per point), written as wave-coalesced float4 row reads.
#include "common.hpp"
#include "ev2hands_hip.h"

namespace {

constexpr int ATT_D = 256;

// one 1024-thread workgroup per (window, hand): thread (part, d) sums its quarter of the points
__global__ __launch_bounds__(1024) void attn_sim_kernel(const float4* __restrict__ logits, const float* __restrict__ query,
                                                        int ldq, size_t hand_stride, int N, float* __restrict__ sim) {
    __shared__ float red[4][4][ATT_D];
    const int b = blockIdx.x, h = blockIdx.y;
    const int d = threadIdx.x & (ATT_D - 1), part = threadIdx.x >> 8;
    const int per = (N + 3) / 4;
    const int n_lo = part * per, n_hi = min(N, n_lo + per);
    const float* q = query + h * hand_stride + (size_t)b * N * ldq + d;
    const float4* kk = logits + (size_t)b * N;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int n = n_lo;
    for (; n + 4 <= n_hi; n += 4) {
        const float q0 = q[(size_t)n * ldq], q1 = q[(size_t)(n + 1) * ldq], q2 = q[(size_t)(n + 2) * ldq],
                    q3 = q[(size_t)(n + 3) * ldq];
        const float4 k0 = kk[n], k1 = kk[n + 1], k2 = kk[n + 2], k3 = kk[n + 3];
        a0 = fmaf(k0.x, q0, a0); a1 = fmaf(k0.y, q0, a1); a2 = fmaf(k0.z, q0, a2); a3 = fmaf(k0.w, q0, a3);
        a0 = fmaf(k1.x, q1, a0); a1 = fmaf(k1.y, q1, a1); a2 = fmaf(k1.z, q1, a2); a3 = fmaf(k1.w, q1, a3);
        a0 = fmaf(k2.x, q2, a0); a1 = fmaf(k2.y, q2, a1); a2 = fmaf(k2.z, q2, a2); a3 = fmaf(k2.w, q2, a3);
        a0 = fmaf(k3.x, q3, a0); a1 = fmaf(k3.y, q3, a1); a2 = fmaf(k3.z, q3, a2); a3 = fmaf(k3.w, q3, a3);
    }
    for (; n < n_hi; ++n) {
        const float q0 = q[(size_t)n * ldq];
        const float4 k0 = kk[n];
        a0 = fmaf(k0.x, q0, a0); a1 = fmaf(k0.y, q0, a1); a2 = fmaf(k0.z, q0, a2); a3 = fmaf(k0.w, q0, a3);
    }
    red[part][0][d] = a0; red[part][1][d] = a1; red[part][2][d] = a2; red[part][3][d] = a3;
    __syncthreads();
    if (part == 0) {
        float s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float t = ((red[0][c][d] + red[1][c][d]) + red[2][c][d]) + red[3][c][d];
            s[c] = 0.0625f * t;     // (value channels = 256) ** -0.5, TEHNet.py:22
        }
        const float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        float e[4], sum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { e[c] = expf(s[c] - mx); sum += e[c]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) sim[(((size_t)b * 2 + h) * 4 + c) * ATT_D + d] = e[c] / sum;
    }
}

// ---- the same similarity map with the LAST query convolution folded behind the reduction over the points.
// query = Conv1d(k=3) -> ReLU -> BN -> Conv1d(k=3) -> BN (TEHNet.py:150-166) enters the attention only through
// sum_n key[c][n] query[d][n], and its last Conv1d -> BN is affine:  query[d][n] = sum_t sum_i W[d][t][i] q1[i][n + t - 1] + b[d]
// (q1 = output of the first conv block, zero padded), hence
//     sum_n key[c][n] query[d][n] = sum_t sum_i W[d][t][i] Z[c][t][i] + b[d] K[c],
//     Z[c][t][i] = sum_n key[c][n] q1[i][n + t - 1],      K[c] = sum_n key[c][n]:
// a [4 x 768] x [768 x 256] product per (window, hand) instead of an [N x 768] x [768 x 256] convolution -- 206 GMAC per
// 256-window batch (both hands) become one streaming pass over q1 and 0.4 GMAC.  Everything in fp32 fma chains with a fixed
// summation order (chunk partials are summed in chunk order: no atomics, results are reproducible and batch-independent).
constexpr int ZS_ROWS = 256;      // points per partial sum

// grid (chunks, B): thread j owns columns 2j, 2j+1 of the 512-wide q1 rows (both hands side by side) and their 12 (class, tap) sums
__global__ __launch_bounds__(256) void attn_zsum_kernel(const float4* __restrict__ logits, const float* __restrict__ q1, int ldq, int N,
                                                        float* __restrict__ zpart) {
    __shared__ float4 sk[ZS_ROWS + 2];        // key rows n0 - 1 .. n0 + ZS_ROWS (zero outside the window)
    const int b = blockIdx.y, ch = blockIdx.x, nch = gridDim.x, tid = threadIdx.x;
    const int n0 = ch * ZS_ROWS, n1 = min(N, n0 + ZS_ROWS);
    for (int i = tid; i < ZS_ROWS + 2; i += 256) {
        const int n = n0 - 1 + i;
        sk[i] = (n >= 0 && n < N) ? logits[(size_t)b * N + n] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    float ax[3][4], ay[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) { ax[t][c] = 0.f; ay[t][c] = 0.f; }
    const float* qp = q1 + ((size_t)b * N + n0) * ldq + 2 * tid;
    auto step = [&](int i, float2 v) {        // row n0 + i: tap t pairs it with key[n0 + i - t + 1] = sk[i + 2 - t]
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float4 k = sk[i + 2 - t];
            ax[t][0] = fmaf(k.x, v.x, ax[t][0]); ax[t][1] = fmaf(k.y, v.x, ax[t][1]); ax[t][2] = fmaf(k.z, v.x, ax[t][2]); ax[t][3] = fmaf(k.w, v.x, ax[t][3]);
            ay[t][0] = fmaf(k.x, v.y, ay[t][0]); ay[t][1] = fmaf(k.y, v.y, ay[t][1]); ay[t][2] = fmaf(k.z, v.y, ay[t][2]); ay[t][3] = fmaf(k.w, v.y, ay[t][3]);
        }
    };
    const int rows = n1 - n0;
    int i = 0;
    for (; i + 4 <= rows; i += 4) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2*>(qp + (size_t)(i + u) * ldq);
#pragma unroll
        for (int u = 0; u < 4; ++u) step(i + u, v[u]);
    }
    for (; i < rows; ++i) step(i, *reinterpret_cast<const float2*>(qp + (size_t)i * ldq));
    float* zp = zpart + ((size_t)b * nch + ch) * 12 * 512 + 2 * tid;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<float2*>(zp + (c * 3 + t) * 512) = make_float2(ax[t][c], ay[t][c]);
}

// grid (B, 2), 1024 threads: thread (q, d) sums the q-th quarter of the 768 contraction steps of column d; the four partial sums
// are added in quarter order
__global__ __launch_bounds__(1024) void attn_simfold_kernel(const float* __restrict__ zpart, int nch, const float4* __restrict__ logits, int N,
                                                            const float* __restrict__ w4t0, const float* __restrict__ w4t1,
                                                            const float* __restrict__ b40, const float* __restrict__ b41, float* __restrict__ sim) {
    __shared__ float4 zs[3 * ATT_D];          // zs[t * 256 + i] = Z[0..3][t][i]
    __shared__ float4 kp[1024];
    __shared__ float4 kp2[32];
    __shared__ float red[3][4][ATT_D];
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, d = tid & (ATT_D - 1), q = tid >> 8;
    const float* w4t = h ? w4t1 : w4t0;
    if (q < 3) {                              // chunk partials of tap q, in chunk order
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < nch; ++ch)
#pragma unroll
            for (int c = 0; c < 4; ++c) z[c] += zpart[(((size_t)b * nch + ch) * 12 + c * 3 + q) * 512 + h * ATT_D + d];
        zs[q * ATT_D + d] = make_float4(z[0], z[1], z[2], z[3]);
    }
    float4 ks = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = tid; n < N; n += 1024) {
        const float4 k = logits[(size_t)b * N + n];
        ks.x += k.x; ks.y += k.y; ks.z += k.z; ks.w += k.w;
    }
    kp[tid] = ks;
    __syncthreads();
    if (tid < 32) {
        float4 k2 = kp[tid * 32];
        for (int j = 1; j < 32; ++j) { const float4 k = kp[tid * 32 + j]; k2.x += k.x; k2.y += k.y; k2.z += k.z; k2.w += k.w; }
        kp2[tid] = k2;
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (q == 0) {
        float4 K = kp2[0];
        for (int j = 1; j < 32; ++j) { const float4 k = kp2[j]; K.x += k.x; K.y += k.y; K.z += k.z; K.w += k.w; }
        const float bd = (h ? b41 : b40)[d];
        a0 = bd * K.x; a1 = bd * K.y; a2 = bd * K.z; a3 = bd * K.w;
    }
    constexpr int KQ = 3 * ATT_D / 4;
#pragma unroll 8
    for (int k = q * KQ; k < (q + 1) * KQ; ++k) {
        const float w = w4t[(size_t)k * ATT_D + d];
        const float4 z = zs[k];
        a0 = fmaf(w, z.x, a0); a1 = fmaf(w, z.y, a1); a2 = fmaf(w, z.z, a2); a3 = fmaf(w, z.w, a3);
    }
    if (q > 0) { red[q - 1][0][d] = a0; red[q - 1][1][d] = a1; red[q - 1][2][d] = a2; red[q - 1][3][d] = a3; }
    __syncthreads();
    if (q == 0) {
        const float t[4] = {a0, a1, a2, a3};
        float s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = 0.0625f * (((t[c] + red[0][c][d]) + red[1][c][d]) + red[2][c][d]);    // 256 ** -0.5, TEHNet.py:22
        const float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        float e[4], sum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { e[c] = expf(s[c] - mx); sum += e[c]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) sim[(((size_t)b * 2 + h) * 4 + c) * ATT_D + d] = e[c] / sum;
    }
}

// one wave per point: lanes hold 4 channels each of the 256-wide value row, both hands' sim in registers
constexpr int CTX_PTS_PER_WAVE = 32;
// VBF16: the value rows are bf16 (the BF16 mode stores l0 that way, forward.hip); ldv counts values either way
// VBF16: 0 = float32 value rows, 1 = bf16 rows (BF16 mode), 2 [r6] = fp16 rows stored times the per-window power of two vscale[b] (F16 mode)
template <int VBF16>
__global__ __launch_bounds__(256) void attn_context_kernel(const float* __restrict__ sim, const float* __restrict__ value,
                                                           int ldv, int N, size_t rows_total, float* __restrict__ hf8,
                                                           unsigned* __restrict__ amax, int amax_hand_stride,
                                                           const float* __restrict__ value_unscale, const float* __restrict__ vscale = nullptr) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 w[2][4];
    // value channel d enters as value[d] * value_unscale[d]: folded into this lane's sim weights once (powers of two: exact)
    const float4 vu = value_unscale ? *reinterpret_cast<const float4*>(value_unscale + lane * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 t = *reinterpret_cast<const float4*>(sim + (((size_t)b * 2 + h) * 4 + c) * ATT_D + lane * 4);
            t.x *= vu.x; t.y *= vu.y; t.z *= vu.z; t.w *= vu.w;
            if constexpr (VBF16 == 2) {                   // the rows' power of two is undone on the weights, once per workgroup (exact)
                const float iv = 1.f / vscale[b];
                t.x *= iv; t.y *= iv; t.z *= iv; t.w *= iv;
            }
            w[h][c] = t;
        }
    const int n0 = (blockIdx.x * 4 + wave) * CTX_PTS_PER_WAVE;
    unsigned am = 0u;
    for (int n = n0; n < n0 + CTX_PTS_PER_WAVE && n < N; ++n) {
        const size_t row = (size_t)b * N + n;
        float4 v;
        if constexpr (VBF16 == 1) {
            const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(value) + row * ldv + lane * 4);
            v = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
        } else if constexpr (VBF16 == 2) {
            typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
            const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(value) + row * ldv + lane * 4);
            const h16x2 a = __builtin_bit_cast(h16x2, h.x), b_ = __builtin_bit_cast(h16x2, h.y);
            v = make_float4((float)a[0], (float)a[1], (float)b_[0], (float)b_[1]);
        } else {
            v = *reinterpret_cast<const float4*>(value + row * ldv + lane * 4);
        }
        float acc[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[h * 4 + c] = fmaf(w[h][c].w, v.w, fmaf(w[h][c].z, v.z, fmaf(w[h][c].y, v.y, w[h][c].x * v.x)));
        // 8 wave-wide sums with 10 shuffles instead of 48: every butterfly step halves the number of values a lane carries
        // (lanes with the step's bit set keep the upper half of the values), then three plain steps finish the 8-lane groups
        float r4[4], r2[2];
        const bool hi32 = lane & 32, hi16 = lane & 16, hi8 = lane & 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) r4[i] = (hi32 ? acc[4 + i] : acc[i]) + __shfl_xor(hi32 ? acc[i] : acc[4 + i], 32, 64);
#pragma unroll
        for (int i = 0; i < 2; ++i) r2[i] = (hi16 ? r4[2 + i] : r4[i]) + __shfl_xor(hi16 ? r4[i] : r4[2 + i], 16, 64);
        float r = (hi8 ? r2[1] : r2[0]) + __shfl_xor(hi8 ? r2[0] : r2[1], 8, 64);
        r += __shfl_xor(r, 4, 64);
        r += __shfl_xor(r, 2, 64);
        r += __shfl_xor(r, 1, 64);
        // lanes 8g..8g+7 now hold sum number g = 4*hi32 + 2*hi16 + hi8 = (hand, class); hf8 row layout [hand][row][8]
        if ((lane & 7) == 0) {
            const int gsum = lane >> 3;
            hf8[((size_t)(gsum >> 2) * rows_total + row) * 8 + (gsum & 3)] = r;
            am = max(am, __float_as_uint(r) & 0x7fffffffu);
        } else if ((lane & 7) == 4) {
            const int gsum = lane >> 3;
            hf8[((size_t)(gsum >> 2) * rows_total + row) * 8 + 4 + (gsum & 3)] = 0.f;
        }
    }
    if (amax) {                          // range records of the two hands' context features (f16x2): lanes 0..31 hold hand 0
        unsigned a0 = wave_max_u32_dpp(lane < 32 ? am : 0u), a1 = wave_max_u32_dpp(lane < 32 ? 0u : am);
        if (lane == 0) {
            if (a0) atomicMax(&amax[b], a0);
            if (a1) atomicMax(&amax[amax_hand_stride + b], a1);
        }
    }
}

}  // namespace

extern "C" int ev2h_attn_sim(const float* logits_pm, const float* query_pm, int ldq, size_t query_hand_stride, int B, int N,
                             float* sim, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(logits_pm && query_pm && sim && B > 0 && N > 0 && ldq >= ATT_D);
    dim3 grid(B, 2);
    attn_sim_kernel<<<grid, 1024, 0, (hipStream_t)stream>>>((const float4*)logits_pm, query_pm, ldq, query_hand_stride, N, sim);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" size_t ev2h_attn_sim_folded_scratch(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * ceil_div(N, ZS_ROWS) * 12 * 512;
}

// Second half of ev2h_attn_sim_folded alone, for partials that the first query convolution's own epilogue produced
// (gemm_bf16.hip: zsum_epilogue; rows_per_partial = 128 there): zpart [B][N / rows_per_partial][12][512].  Internal (forward.hip).
int ev2h_attn_simfold_partials(const float* zpart, int rows_per_partial, const float* logits_pm, int B, int N, const float* w4t_left,
                               const float* w4t_right, const float* b4_left, const float* b4_right, float* sim, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(zpart && logits_pm && w4t_left && w4t_right && b4_left && b4_right && sim && B > 0 && N > 0 && rows_per_partial > 0 &&
                   N % rows_per_partial == 0);
    attn_simfold_kernel<<<dim3(B, 2), 1024, 0, (hipStream_t)stream>>>(zpart, N / rows_per_partial, (const float4*)logits_pm, N, w4t_left, w4t_right,
                                                                      b4_left, b4_right, sim);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" int ev2h_attn_sim_folded(const float* logits_pm, const float* q1_pm, int ldq, int B, int N, const float* w4t_left,
                                    const float* w4t_right, const float* b4_left, const float* b4_right, float* scratch, float* sim,
                                    ev2h_stream_t stream) {
    EV2H_CHECK_ARG(logits_pm && q1_pm && w4t_left && w4t_right && b4_left && b4_right && scratch && sim);
    EV2H_CHECK_ARG(B > 0 && N > 0 && ldq >= 2 * ATT_D && (ldq % 2) == 0);
    const int nch = ceil_div(N, ZS_ROWS);
    attn_zsum_kernel<<<dim3(nch, B), 256, 0, (hipStream_t)stream>>>((const float4*)logits_pm, q1_pm, ldq, N, scratch);
    EV2H_CHECK_LAUNCH();
    attn_simfold_kernel<<<dim3(B, 2), 1024, 0, (hipStream_t)stream>>>(scratch, nch, (const float4*)logits_pm, N, w4t_left, w4t_right, b4_left,
                                                                      b4_right, sim);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" int ev2h_attn_context(const float* sim, const float* value_pm, int ldv, int B, int N, float* hf8, uint32_t* hf_amax,
                                 int amax_hand_stride, const float* value_unscale, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(sim && value_pm && hf8 && B > 0 && N > 0 && ldv >= ATT_D && (ldv % 4) == 0);
    dim3 grid(ceil_div(N, 4 * CTX_PTS_PER_WAVE), B);
    attn_context_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(sim, value_pm, ldv, N, (size_t)B * N, hf8, hf_amax, amax_hand_stride, value_unscale);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

// internal (forward.hip, BF16 mode): the same with bf16 value rows
int ev2h_attn_context_bf16rows(const float* sim, const void* value_pm, int ldv, int B, int N, float* hf8, const float* value_unscale, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(sim && value_pm && hf8 && B > 0 && N > 0 && ldv >= ATT_D && (ldv % 4) == 0);
    dim3 grid(ceil_div(N, 4 * CTX_PTS_PER_WAVE), B);
    attn_context_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(sim, reinterpret_cast<const float*>(value_pm), ldv, N, (size_t)B * N, hf8, nullptr, 0, value_unscale);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

// internal (forward.hip, F16 mode) [r6]: fp16 value rows stored times the per-window power of two vscale[b]; with the range records of
// the two hands' context features (the F16 regressors scale by them)
int ev2h_attn_context_f16rows(const float* sim, const void* value_pm, int ldv, int B, int N, float* hf8, uint32_t* hf_amax, int amax_hand_stride,
                              const float* value_unscale, const float* vscale, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(sim && value_pm && hf8 && vscale && B > 0 && N > 0 && ldv >= ATT_D && (ldv % 4) == 0);
    dim3 grid(ceil_div(N, 4 * CTX_PTS_PER_WAVE), B);
    attn_context_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(sim, reinterpret_cast<const float*>(value_pm), ldv, N, (size_t)B * N, hf8, hf_amax, amax_hand_stride,
                                                                 value_unscale, vscale);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}
