// Fused grouped set-abstraction MLP for gfx950: gather -> layer-1 finish -> layer 2 -> layer 3 -> max over
// the K neighbours of each centroid.  Reference: PointNetSetAbstractionMsg.forward,
// /root/reference/src/Ev2Hands/model/pointnet2_utils.py:241-257 (gather, centre, concat
// [features, rel-xyz], 3 x (1x1 Conv2d -> BN -> ReLU), max over K).  77 % of the path's MACs.
//
// MI355X design (not a translation of the eager graph):
//  * layer 1 is linear in [features(idx), xyz(idx) - centre], so its feature part is a per-POINT
//    table P1 = W1f' f + b1' computed once per point by the dense GEMM (not once per (centroid,
//    neighbour) row); this kernel gathers P1 rows and adds the 3-term relative-xyz part exactly
//    (dx computed as in the reference), then ReLU.  BN is folded into W/b by the host.
//  * one wavefront owns a 32-neighbour strip.  Activations never touch LDS or HBM:
//      layer 2:  D2[channel][neighbour] = W2tile(A, from LDS) x H1(B, registers)
//      layer 3:  D3[neighbour][channel] = H2(A = D2 registers, unchanged) x W3tile(B, from LDS)
//    v_mfma_f32_32x32x2_f32's D layout (lane = column, 16 rows in registers) is exactly the
//    A-operand layout of the next MFMA when the k-slot order of W3 follows mfma_row(), so the
//    layer-2 result is consumed in place; the max over neighbours is an in-register max over the
//    16 D3 rows + one cross-half shuffle.
//  * weight tiles (layer 2: all outputs x 32 inputs; layer 3: 32 outputs x all inputs) stream through a
//    double-buffered LDS tile shared by the 8 waves of the workgroup; the next tile is prefetched into registers while
//    the current one is multiplied -> one barrier per tile.
#include "common.hpp"
#include "ev2hands_hip.h"

namespace {

struct SaP {
    const float* P1; int ldp;
    const float4* pts4;
    const float4* ctr4;
    const int32_t* gidx;
    const float4* W1x;
    const float* W2; const float* b2;
    const float* W3; const float* b3;
    float* out; int ldo;
    int B, Npts, S, K;
    int nblk;
    float* xyz_out; int xyz_ld;          // optional: the group's centroid as 8 more columns of the consumer's input rows (ev2h_sa_desc)
};

constexpr int SA_WAVES = 8;
constexpr int SA_THREADS = SA_WAVES * 64;

template <int C1, int C2, int C3>
struct SaCfg {
    static constexpr int T2 = (C2 + 31) / 32;          // layer-2 output tiles (rows of W2 padded to T2*32)
    static constexpr int C2K = (C2 + 7) / 8 * 8;       // layer-3 contraction length (W3 columns, zero padded)
    static constexpr int T3 = C3 / 32;
    static constexpr int NC1 = C1 / 32;                // layer-2 contraction chunks of 32 input channels
    static constexpr int LD2 = 32 + 4;                 // W2 chunk tile: [T2*32 rows][32 cols], padded stride
    static constexpr int LD3 = C2K + 4;                // W3 tile: [32 rows][C2K cols], padded stride
    static constexpr int TILE2 = T2 * 32 * LD2;
    static constexpr int TILE3 = 32 * LD3;
    static constexpr int TILE = (TILE2 > TILE3 ? TILE2 : TILE3);
    static constexpr int N2V = T2 * 32 * 8;            // float4 per W2 chunk tile
    static constexpr int N3V = 32 * C2K / 4;           // float4 per W3 tile
    static constexpr int NV = (N2V > N3V ? N2V : N3V);
    static constexpr int NLD = (NV + SA_THREADS - 1) / SA_THREADS;
    static constexpr int LDS_FLOATS = 2 * TILE + C1 * 4 + T2 * 32;
    static constexpr int REM = C2 % 32;
    static constexpr int NQ_LAST = REM ? (REM + 7) / 8 : 4;   // live 8-channel blocks of the last layer-2 tile
    // prefetch the next 32-channel chunk of gathered P1 rows one chunk ahead, except where the 7x16
    // accumulator registers leave no room (the second wave on the SIMD hides that latency instead)
    static constexpr bool PREFETCH_P1 = (T2 < 7);
};

template <int C1, int C2, int C3>
__global__ __launch_bounds__(SA_THREADS, 2) void sa_mlp_max_kernel(SaP p) {
    using Cfg = SaCfg<C1, C2, C3>;
    constexpr int T2 = Cfg::T2, C2K = Cfg::C2K, T3 = Cfg::T3, NC1 = Cfg::NC1, LD2 = Cfg::LD2, LD3 = Cfg::LD3;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    float* wt0 = smem;
    float* wt1 = smem + Cfg::TILE;
    float4* sW1x = reinterpret_cast<float4*>(smem + 2 * Cfg::TILE);
    float* sb2 = smem + 2 * Cfg::TILE + C1 * 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int L = xcd_remap(blockIdx.x, p.nblk);
    const int ngroups = p.B * p.S;
    const int g = L * SA_WAVES + wave;
    const bool valid = g < ngroups;
    const int gg = valid ? g : ngroups - 1;
    const int b = gg / p.S;

    for (int i = tid; i < C1; i += SA_THREADS) sW1x[i] = p.W1x[i];
    for (int i = tid; i < T2 * 32; i += SA_THREADS) sb2[i] = p.b2[i];

    // ---- weight-tile staging (global -> registers -> LDS)
    //   W2 chunk tile c: all T2*32 output rows x input channels [32c, 32c+32)
    //   W3 tile u:       output rows [32u, 32u+32) x all C2K inputs
    // (loads are unconditional on a clamped element index so `stg` stays in registers)
    f32x4 stg[Cfg::NLD];
    auto load_w2 = [&](int c) {
#pragma unroll
        for (int i = 0; i < Cfg::NLD; ++i) {
            int e = tid + i * SA_THREADS;
            e = e < Cfg::N2V ? e : Cfg::N2V - 1;
            stg[i] = *reinterpret_cast<const f32x4*>(p.W2 + (size_t)(e >> 3) * C1 + 32 * c + (e & 7) * 4);
        }
    };
    auto store_w2 = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < Cfg::NLD; ++i) {
            const int e = tid + i * SA_THREADS;
            if (e < Cfg::N2V) *reinterpret_cast<f32x4*>(dst + (e >> 3) * LD2 + (e & 7) * 4) = stg[i];
        }
    };
    auto load_w3 = [&](int u) {
#pragma unroll
        for (int i = 0; i < Cfg::NLD; ++i) {
            int e = tid + i * SA_THREADS;
            e = e < Cfg::N3V ? e : Cfg::N3V - 1;
            stg[i] = *reinterpret_cast<const f32x4*>(p.W3 + (size_t)(32 * u + e / (C2K / 4)) * C2K + (e % (C2K / 4)) * 4);
        }
    };
    auto store_w3 = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < Cfg::NLD; ++i) {
            const int e = tid + i * SA_THREADS;
            if (e < Cfg::N3V) *reinterpret_cast<f32x4*>(dst + (e / (C2K / 4)) * LD3 + (e % (C2K / 4)) * 4) = stg[i];
        }
    };

    float mrun[T3];
#pragma unroll
    for (int u = 0; u < T3; ++u) mrun[u] = -INFINITY;

    const float4 ctr = p.ctr4[gg];
    const int nstrips = p.K >> 5;
    const int32_t* gi = p.gidx + (size_t)gg * p.K;

    int buf = 0;   // parity of the running tile counter selects the LDS buffer; (NC1 + T3) tiles per strip
    load_w2(0);
    store_w2(wt0);
    __syncthreads();

    for (int strip = 0; strip < nstrips; ++strip) {
        const int idx = gi[strip * 32 + l31];
        const float4 q = p.pts4[(size_t)b * p.Npts + idx];
        const float dx = __fsub_rn(q.x, ctr.x), dy = __fsub_rn(q.y, ctr.y), dz = __fsub_rn(q.z, ctr.z);
        // this lane's layer-1 channels of chunk c: 32c + 16*half + [0,16)
        const float4* prow = reinterpret_cast<const float4*>(p.P1 + ((size_t)b * p.Npts + idx) * p.ldp + 16 * half);
        float4 raw[4];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) raw[j4] = prow[j4];

        // ---------------- layer 2, contraction-chunk outer: h2[t] accumulates D2[channel 32t + mfma_row(r,half)][neighbour]
        f32x16 h2[T2];
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h2[t][r] = 0.f;

#pragma unroll 1
        for (int c = 0; c < NC1; ++c) {
            float* cur = buf ? wt1 : wt0;
            float* nxt = buf ? wt0 : wt1;
            if (!Cfg::PREFETCH_P1 && c > 0) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) raw[j4] = prow[c * 8 + j4];
            }
            if (c + 1 < NC1) load_w2(c + 1); else load_w3(0);
            // layer-1 finish: relu(P1 + W1x . (xyz[idx] - ctr)), exact relative coordinates as in the reference
            float h1[16];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const float vv[4] = {raw[j4].x, raw[j4].y, raw[j4].z, raw[j4].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float4 w = sW1x[32 * c + 16 * half + j4 * 4 + e];
                    const float t = __fmaf_rn(w.z, dz, __fmaf_rn(w.y, dy, __fmaf_rn(w.x, dx, vv[e])));
                    h1[j4 * 4 + e] = fmaxf(t, 0.f);
                }
            }
            if (Cfg::PREFETCH_P1 && c + 1 < NC1) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) raw[j4] = prow[(c + 1) * 8 + j4];
            }
            const float* pa = cur + l31 * LD2 + 16 * half;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                // 32x32x2 f32 MFMA: issue interval == dependent latency (64 cycles), so 4 back-to-back
                // MFMAs on one accumulator lose nothing and keep a single A fragment live
#pragma unroll
                for (int t = 0; t < T2; ++t) {
                    const float4 a = *reinterpret_cast<const float4*>(pa + 32 * t * LD2 + j4 * 4);
                    h2[t] = mfma32(a.x, h1[j4 * 4 + 0], h2[t]);
                    h2[t] = mfma32(a.y, h1[j4 * 4 + 1], h2[t]);
                    h2[t] = mfma32(a.z, h1[j4 * 4 + 2], h2[t]);
                    h2[t] = mfma32(a.w, h1[j4 * 4 + 3], h2[t]);
                }
            }
            if (c + 1 < NC1) store_w2(nxt); else store_w3(nxt);
            __syncthreads();
            buf ^= 1;
        }
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h2[t][r] = fmaxf(h2[t][r] + sb2[32 * t + mfma_row(r, half)], 0.f);

        // ---------------- layer 3 + max over the strip's 32 neighbours: D3[neighbour][channel] = H2 (A, registers) x W3 tile (B)
#pragma unroll 1
        for (int u = 0; u < T3; ++u) {
            float* cur = buf ? wt1 : wt0;
            float* nxt = buf ? wt0 : wt1;
            const bool more_w3 = (u + 1 < T3);
            const bool more = more_w3 || (strip + 1 < nstrips);
            if (more_w3) load_w3(u + 1); else if (more) load_w2(0);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* pb = cur + l31 * LD3 + half * 4;
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                const int nq = (t == T2 - 1) ? Cfg::NQ_LAST : 4;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    if (qq < nq) {
                        const float4 w = *reinterpret_cast<const float4*>(pb + 32 * t + 8 * qq);
                        acc = mfma32(h2[t][4 * qq + 0], w.x, acc);
                        acc = mfma32(h2[t][4 * qq + 1], w.y, acc);
                        acc = mfma32(h2[t][4 * qq + 2], w.z, acc);
                        acc = mfma32(h2[t][4 * qq + 3], w.w, acc);
                    }
                }
            }
            float m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
#pragma unroll
            for (int uu = 0; uu < T3; ++uu) mrun[uu] = (uu == u) ? fmaxf(mrun[uu], m) : mrun[uu];
            if (more_w3) store_w3(nxt); else if (more) store_w2(nxt);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---------------- bias + ReLU commute with the max (both monotone): out = relu(max_k acc + b3)
#pragma unroll
    for (int u = 0; u < T3; ++u) {
        const float v = fmaxf(mrun[u], __shfl_xor(mrun[u], 32, 64));
        if (valid && half == 0) p.out[(size_t)g * p.ldo + 32 * u + l31] = fmaxf(v + p.b3[32 * u + l31], 0.f);
    }
    if (p.xyz_out && valid && lane < 8) {
        const float4 c = p.ctr4[g];
        p.xyz_out[(size_t)g * p.xyz_ld + lane] = lane == 0 ? c.x : lane == 1 ? c.y : lane == 2 ? c.z : 0.f;
    }
}

template <int C1, int C2, int C3>
int launch_sa(const SaP& p, hipStream_t st) {
    using Cfg = SaCfg<C1, C2, C3>;
    static PerDevice attr_set{};
    const int lds = Cfg::LDS_FLOATS * (int)sizeof(float);
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sa_mlp_max_kernel<C1, C2, C3>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds)););
    sa_mlp_max_kernel<C1, C2, C3><<<p.nblk, SA_THREADS, lds, st>>>(p);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

}  // namespace

int ev2h_sa_mlp_max_bf16(const ev2h_sa_desc* d, ev2h_stream_t stream);

extern "C" int ev2h_sa_mlp_max(const ev2h_sa_desc* d, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(d && (d->P1 || d->feat) && d->pts4 && d->ctr4 && d->gidx && d->W1x && d->b2 && d->b3 && d->out);
    EV2H_CHECK_ARG(d->B > 0 && d->S > 0 && d->Npts > 0 && d->K >= 32 && (d->K % 32) == 0);
    EV2H_CHECK_ARG((d->ldp % 4) == 0);
    EV2H_CHECK_ARG(!d->xyz_out || (d->xyz_ld >= 8 && (d->xyz_ld % 4) == 0));
    if (d->precision != EV2H_PREC_F32) return ev2h_sa_mlp_max_bf16(d, stream);
    EV2H_CHECK_ARG(d->s_off == 0 && (d->S_total == 0 || d->S_total == d->S));      // (centroid sub-ranges: the plane-mode kernels only)
    // exact fp32: layer 1 is the gathered table row + the relative-coordinate term; the raw-feature form (ev2h_sa_desc.feat
    // without a table) exists on the matrix pipe only
    if (!d->P1) {
        ev2h_set_error("ev2h_sa_mlp_max: EV2H_PREC_F32 needs the layer-1 table P1 (feature rows without a table: BF16 / F16X2 only)");
        return EV2H_ERR_ARG;
    }
    EV2H_CHECK_ARG(d->W2 && d->W3);
    SaP p{};
    p.P1 = d->P1; p.ldp = d->ldp; p.pts4 = (const float4*)d->pts4; p.ctr4 = (const float4*)d->ctr4; p.gidx = d->gidx;
    p.W1x = (const float4*)d->W1x; p.W2 = d->W2; p.b2 = d->b2; p.W3 = d->W3; p.b3 = d->b3;
    p.out = d->out; p.ldo = d->ldo; p.B = d->B; p.Npts = d->Npts; p.S = d->S; p.K = d->K;
    p.nblk = ceil_div(d->B * d->S, SA_WAVES);
    p.xyz_out = d->xyz_out; p.xyz_ld = d->xyz_ld;
    hipStream_t st = (hipStream_t)stream;
    const int c1 = d->C1, c2 = d->C2, c3 = d->C3;
    if (c1 == 32 && c2 == 32 && c3 == 64) return launch_sa<32, 32, 64>(p, st);
    if (c1 == 64 && c2 == 64 && c3 == 128) return launch_sa<64, 64, 128>(p, st);
    if (c1 == 64 && c2 == 96 && c3 == 128) return launch_sa<64, 96, 128>(p, st);
    if (c1 == 128 && c2 == 128 && c3 == 256) return launch_sa<128, 128, 256>(p, st);
    if (c1 == 128 && c2 == 196 && c3 == 256) return launch_sa<128, 196, 256>(p, st);
    ev2h_set_error("ev2h_sa_mlp_max: unsupported MLP widths %d-%d-%d", c1, c2, c3);
    return EV2H_ERR_ARG;
}
