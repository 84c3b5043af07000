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

// one wave per point: lanes hold 4 channels each of the 256-wide value row, both hands' sim in registers
constexpr int CTX_PTS_PER_WAVE = 32;
__global__ __launch_bounds__(256) void attn_context_kernel(const float* __restrict__ sim, const float* __restrict__ value,
                                                           int ldv, int N, size_t rows_total, float* __restrict__ hf8,
                                                           unsigned* __restrict__ amax, int amax_hand_stride) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 w[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            w[h][c] = *reinterpret_cast<const float4*>(sim + (((size_t)b * 2 + h) * 4 + c) * ATT_D + lane * 4);
    const int n0 = (blockIdx.x * 4 + wave) * CTX_PTS_PER_WAVE;
    unsigned am = 0u;
    for (int n = n0; n < n0 + CTX_PTS_PER_WAVE && n < N; ++n) {
        const size_t row = (size_t)b * N + n;
        const float4 v = *reinterpret_cast<const float4*>(value + row * ldv + lane * 4);
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

extern "C" int ev2h_attn_context(const float* sim, const float* value_pm, int ldv, int B, int N, float* hf8, uint32_t* hf_amax,
                                 int amax_hand_stride, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(sim && value_pm && hf8 && B > 0 && N > 0 && ldv >= ATT_D && (ldv % 4) == 0);
    dim3 grid(ceil_div(N, 4 * CTX_PTS_PER_WAVE), B);
    attn_context_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(sim, value_pm, ldv, N, (size_t)B * N, hf8, hf_amax, amax_hand_stride);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}
