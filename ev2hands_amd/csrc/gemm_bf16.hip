// Dense layers on the 16-bit matrix pipe: same contract as gemm.hip (ev2h_gemm), operands split on the fly into the
// NS planes of planes.hpp (NS = 2 "f16x2", NS = 3 "bf16x3": fp32-class; NS = 1 plain bf16; NS = 4 [r6] the mode code of "f16": ONE
// fp16 plane with f16x2's range scaling -- plane_count(NS) planes are stored, planes_f16(NS) selects the scaling), fp32 accumulate.
// Three kernels: a generic one that splits both operands (W without a plane image: the tiny heads), a wide 128x256
// one and the default 128x128 "occupancy" kernel, both with host-packed W plane images streamed by LDS-DMA.
// First kernel:
// 128x128x32 tiles, 8 waves (2 x 4), each wave 64 x 32 outputs (2 accumulator tiles); LDS rows hold the
// NS planes side by side (NS*64 B + 16 B pad => conflict-free ds_read_b128); register prefetch of the
// next K tile, two LDS buffers, one barrier per K tile.
#include <cstdlib>

#include "planes.hpp"
#include "ev2hands_hip.h"

namespace {


constexpr int GB_BM = 128, GB_BN = 128, GB_BK = 32, GB_THREADS = 512;
constexpr int GO_THREADS_ = 256, GB_BN_ = 128;      // (occupancy kernel: threads, tile columns -- used by zsum_epilogue above its definition)

struct GemmBP {
    const float* X; int ldx;
    const float* W; int ldw;
    float* Y; int ldy;
    int M, N, K;
    const float* bias; int bias_group_rows; int ldbias;
    int relu;
    const float* post_scale; const float* post_shift;
    int taps; int Kc; int rows_per_seq;
    int rowmax_rows;
    float w_unscale;        // power of two: the W planes are those of W / w_unscale, products are multiplied back (csrc/pack.hip: plane_unscale)
    // f16x2 activation range (planes.hpp header; all optional, ignored by the other modes)
    const unsigned* x_amax; const unsigned* x_amax2; int x_group_rows;   // max|X| per group of x_group_rows rows (max of the two sources)
    unsigned* y_amax; int y_group_rows;                                   // out: atomicMax of |Y| per group of output rows
    float* y_scale; float y_bound_w, y_bound_b;                           // out: Y is stored times the power of two that keeps
                                                                          // (y_bound_w * max|X| + y_bound_b) * s below 2^15
    int tiles_n; int nblk;
    // TAP3 occupancy kernel only: the attention's key-weighted column sums (attention.hip: Z[c][t][i]) formed in the epilogue instead of
    // storing Y -- zs_key = logits, point-major [M][4]; zs_out = chunk partials [M / 128][12][N] (see zsum_epilogue)
    const float4* zs_key; float* zs_out;
    int x_bf16;             // TAP3 kernel, NS = 1 / NS = 4 only (internal): X holds bf16 / fp16 values (ldx counts values) -- the rows ARE the operand plane
    const float* x_scale;   // NS = 4 with x_bf16 [r6]: the power of two each group's fp16 rows were stored with (float [groups]); replaces x_amax
};

template <int NS>
struct GBCfg {
    static constexpr int RS = plane_count(NS) * 64 + 16;      // bytes per LDS row (planes of 32 16-bit values + pad)
    static constexpr int OPER = GB_BM * RS;                 // one operand tile
    static constexpr int LDS_BYTES = 4 * OPER;              // 2 buffers x (A, B)
};


// power-of-two scale of X row m (f16x2 with a range record, 1 otherwise)
template <int NS>
__device__ __forceinline__ float x_row_scale(const GemmBP& p, long m) {
    if constexpr (!planes_f16(NS)) return 1.f;
    if (p.x_scale) {                         // fp16 rows stored scaled: that power of two IS the operand's scale
        const long mm = m < 0 ? 0 : (m < p.M ? m : p.M - 1);
        return p.x_scale[mm / p.x_group_rows];
    }
    if (!p.x_amax) return 1.f;
    const long mm = m < 0 ? 0 : (m < p.M ? m : p.M - 1);
    const long g = mm / p.x_group_rows;
    unsigned a = p.x_amax[g];
    if (p.x_amax2) a = max(a, p.x_amax2[g]);
    return f16x2_scale(a);
}
template <int NS>
__device__ __forceinline__ void scale_rows(f32x4& v, float s) {
    if constexpr (planes_f16(NS)) v *= s;       // exact (power of two); the products are multiplied back in the epilogue
}

// Epilogue shared by the three kernels.  acc[i][j][r] = output (row m0 + row0 + 32 i + mfma_row(r, half), column col0 + 32 j + l31):
//   v = acc * (w_unscale / x_scale(row)) + bias;  ReLU;  post affine;  [* y_scale];  store, or max over the tile's 128 rows;
//   optional atomicMax of |v| into the output's range record.
// GEN = false (the two fast kernels, and the generic one on aligned shapes): every 128-row tile lies inside one range group, so
// the x scale, the storage scale and the output record are per tile.  GEN = true (generic kernel only; shapes whose groups
// straddle tiles, e.g. N % 128 != 0, or one row per group: the per-window head layers): per row.
template <int NS, int NI, int NJ, bool GEN>
__device__ __forceinline__ void gemm_epilogue(const GemmBP& p, f32x16 (&acc)[NI][NJ], int m0, int row0, int col0, int wm, int wcol, int tile_cols,
                                              float* red, int tid) {
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const float* bias = p.bias;
    if (bias && p.bias_group_rows > 0) bias += (long)(m0 / p.bias_group_rows) * p.ldbias;
    float bj[NJ], sj[NJ], tj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = col0 + j * 32 + l31;
        const bool okc = col < p.N;
        bj[j] = (bias && okc) ? bias[col] : 0.f;
        sj[j] = (p.post_scale && okc) ? p.post_scale[col] : 1.f;
        tj[j] = (p.post_shift && okc) ? p.post_shift[col] : 0.f;
    }
    const bool xs_on = planes_f16(NS) && p.x_amax != nullptr;
    const bool ys_on = planes_f16(NS) && p.y_scale != nullptr;
    const bool track = planes_f16(NS) && p.y_amax != nullptr;
    auto y_store_scale = [&](long row) {          // power of two that keeps y_bound_w * max|X_g| + y_bound_b below 2^15
        const long rr = row < p.M ? row : p.M - 1;
        const long g = rr / p.x_group_rows;
        unsigned a = p.x_amax[g];
        if (p.x_amax2) a = max(a, p.x_amax2[g]);
        return f16x2_scale(__float_as_uint(__fmaf_rn(p.y_bound_w, __uint_as_float(a), p.y_bound_b)));
    };
    float cx_u = p.w_unscale;
    float sy_u = 1.f;
    if (xs_on) cx_u = p.w_unscale * pow2_inverse(x_row_scale<NS>(p, m0));
    if (ys_on) {
        sy_u = y_store_scale(m0);
        if (!GEN && tid == 0 && col0 == wcol && (m0 % p.y_group_rows) == 0) p.y_scale[m0 / p.y_group_rows] = sy_u;
    }
    unsigned am = 0u;
    float amf = 0.f;
    if (p.rowmax_rows == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + row0 + i * 32 + mfma_row(r, half);
                float cx = cx_u, sy = sy_u;
                if constexpr (GEN) {
                    if (xs_on) cx = p.w_unscale * pow2_inverse(x_row_scale<NS>(p, row));
                    if (ys_on) {
                        sy = y_store_scale(row);
                        if (row < p.M && (row % p.y_group_rows) == 0 && col0 + l31 == 0) p.y_scale[row / p.y_group_rows] = sy;
                    }
                }
                float amr = 0.f;           // |v| maxima as floats: one v_max_f32 with an abs modifier per value
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int col = col0 + j * 32 + l31;
                    float v = acc[i][j][r] * cx + bj[j];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.post_scale) v = __fmaf_rn(v, sj[j], tj[j]);
                    if constexpr (planes_f16(NS)) v *= sy;
                    if (row < p.M && col < p.N) {
                        p.Y[(long)row * p.ldy + col] = v;
                        if constexpr (planes_f16(NS)) amr = fmaxf(amr, fabsf(v));
                    }
                }
                if constexpr (GEN) {
                    if (track) {           // reduce over the 32 columns of the half-wave, one atomic per row
                        unsigned ab = __float_as_uint(amr);
#pragma unroll
                        for (int o = 16; o >= 1; o >>= 1) ab = max(ab, (unsigned)__shfl_xor((int)ab, o, 64));
                        if (l31 == 0 && row < p.M && ab) atomicMax(&p.y_amax[row / p.y_group_rows], ab);
                    }
                } else {
                    amf = fmaxf(amf, amr);
                }
            }
        if (!GEN && track) {
            am = wave_max_u32_dpp(__float_as_uint(amf));
            if (lane == 0 && am) atomicMax(&p.y_amax[m0 / p.y_group_rows], am);
        }
    } else {
        // group-all max over the tile's 128 rows (rowmax_rows == 128: one window per tile; x scale tile-uniform)
        __syncthreads();                 // operand tiles are dead
        float mx[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) mx[j] = -INFINITY;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + row0 + i * 32 + mfma_row(r, half);
                    float v = acc[i][j][r] * cx_u + bj[j];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.post_scale) v = __fmaf_rn(v, sj[j], tj[j]);
                    if (row < p.M) mx[j] = fmaxf(mx[j], v);
                }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], 32, 64));
            if (half == 0) red[wm * tile_cols + wcol + j * 32 + l31] = mx[j];
        }
        __syncthreads();
        if (tid < tile_cols) {
            const int c = col0 - wcol + tid;
            float v = fmaxf(red[tid], red[tile_cols + tid]);
            if (c < p.N) {
                p.Y[(long)(m0 / p.rowmax_rows) * p.ldy + c] = v;
                am = abs_bits(v);
            }
        }
        if (track && tid < tile_cols) {
            am = wave_max_u32_dpp(am);
            if (lane == 0 && am) atomicMax(&p.y_amax[(m0 / p.rowmax_rows) / p.y_group_rows], am);
        }
    }
}

template <int NS, bool GEN>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_nt_bf16_kernel(GemmBP p) {
    using Cfg = GBCfg<NS>;
    constexpr int RS = Cfg::RS;
    const float w_prescale = 1.f / p.w_unscale;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA0 = smem;
    char* sB0 = smem + Cfg::OPER;
    char* sA1 = smem + 2 * Cfg::OPER;
    char* sB1 = smem + 3 * Cfg::OPER;

    const int L = xcd_remap(blockIdx.x, p.nblk);
    const int tn = L % p.tiles_n, tm = L / p.tiles_n;
    const int m0 = tm * GB_BM, n0 = tn * GB_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;

    // loader: thread -> (row = tid / 4, 8 consecutive k = (tid % 4) * 8) of the A tile and of the B tile
    const int lrow = tid >> 2, lseg = tid & 3;
    f32x4 ra[2], rb[2];
    const float xs = x_row_scale<NS>(p, GEN ? m0 + lrow : m0);

    auto gload = [&](int kt) {
        const int k = kt * GB_BK + lseg * 8;
        int tap = 0, kc = k;
        if (p.taps == 3) { tap = k / p.Kc; kc = k - tap * p.Kc; }
        {
            const int m = m0 + lrow;
            bool ok = (m < p.M) && (k < p.K);
            long src = m;
            if (p.taps == 3) {
                const int pos = m % p.rows_per_seq + tap - 1;
                ok = ok && (pos >= 0) && (pos < p.rows_per_seq);
                src = (long)m + tap - 1;
            }
            const long mm = ok ? src : 0;
            const int kk = ok ? kc : 0;
            const f32x4* g = reinterpret_cast<const f32x4*>(p.X + mm * p.ldx + kk);
            f32x4 v0 = g[0], v1 = g[1];
            if (!ok) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
            scale_rows<NS>(v0, xs); scale_rows<NS>(v1, xs);
            ra[0] = v0; ra[1] = v1;
        }
        {
            const int n = n0 + lrow;
            const bool ok = (n < p.N) && (k < p.K);
            const f32x4* g = reinterpret_cast<const f32x4*>(p.W + (long)(ok ? n : 0) * p.ldw + (ok ? k : 0));
            f32x4 v0 = g[0], v1 = g[1];
            if (!ok) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
            rb[0] = v0 * w_prescale; rb[1] = v1 * w_prescale;       // exact (power of two)
        }
    };
    auto swrite_one = [&](char* dst, const f32x4 (&r)[2]) {
        unsigned q[4][plane_count(NS)];
        split_planes<NS>(r[0][0], r[0][1], q[0]);
        split_planes<NS>(r[0][2], r[0][3], q[1]);
        split_planes<NS>(r[1][0], r[1][1], q[2]);
        split_planes<NS>(r[1][2], r[1][3], q[3]);
#pragma unroll
        for (int s = 0; s < plane_count(NS); ++s) {
            u32x4 v = {q[0][s], q[1][s], q[2][s], q[3][s]};
            *reinterpret_cast<u32x4*>(dst + lrow * RS + s * 64 + lseg * 16) = v;
        }
    };

    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    const int nk = (p.K + GB_BK - 1) / GB_BK;
    gload(0);
    swrite_one(sA0, ra);
    swrite_one(sB0, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const char* sA = (kt & 1) ? sA1 : sA0;
        const char* sB = (kt & 1) ? sB1 : sB0;
        if (kt + 1 < nk) gload(kt + 1);
        const char* pa = sA + (wm * 64 + l31) * RS + half * 16;
        const char* pb = sB + (wn * 32 + l31) * RS + half * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            u32x4 b[plane_count(NS)], a0[plane_count(NS)], a1[plane_count(NS)];
#pragma unroll
            for (int s = 0; s < plane_count(NS); ++s) {
                b[s] = *reinterpret_cast<const u32x4*>(pb + s * 64 + m * 32);
                a0[s] = *reinterpret_cast<const u32x4*>(pa + s * 64 + m * 32);
                a1[s] = *reinterpret_cast<const u32x4*>(pa + 32 * RS + s * 64 + m * 32);
            }
#pragma unroll
            for (int q = 0; q < Planes<NS>::NPROD; ++q) {
                acc[0][0] = mfma_planes<NS>(a0[Planes<NS>::A[q]], b[Planes<NS>::B[q]], acc[0][0]);
                acc[1][0] = mfma_planes<NS>(a1[Planes<NS>::A[q]], b[Planes<NS>::B[q]], acc[1][0]);
            }
        }
        if (kt + 1 < nk) {
            swrite_one((kt & 1) ? sA0 : sA1, ra);
            swrite_one((kt & 1) ? sB0 : sB1, rb);
        }
        __syncthreads();
    }
    gemm_epilogue<NS, 2, 1, GEN>(p, acc, m0, wm * 64, n0 + wn * 32, wm, wn * 32, GB_BN, reinterpret_cast<float*>(smem), tid);
}

template <int NS, bool GEN>
int launch_gb(const GemmBP& p, hipStream_t st) {
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_kernel<NS, GEN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GBCfg<NS>::LDS_BYTES)););
    gemm_nt_bf16_kernel<NS, GEN><<<p.nblk, GB_THREADS, GBCfg<NS>::LDS_BYTES, st>>>(p);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

// ---------------------------------------------------------------------------------------- wide tile, pre-split W
// 128 x 256 x 32 tiles for the big layers (N >= 192): W arrives as host-packed bf16 plane images of the LDS
// tile (csrc/pack.hip: gemm_image) and is streamed by LDS-DMA; only X is split on the fly.
// 8 waves as 2 x 4, each wave 64 x 64 = 2 x 2 accumulator tiles: 12 fragment reads feed 24 MFMAs.
constexpr int GW_BN = 256;

template <int NS>
struct GWCfg {
    static constexpr int RS = plane_count(NS) * 64 + 16;
    static constexpr int A_BYTES = GB_BM * RS;
    static constexpr int B_BYTES = GW_BN * RS;
    static constexpr int LDS_BYTES = 2 * (A_BYTES + B_BYTES);
};

template <int NS>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_nt_bf16_wide_kernel(GemmBP p, const char* __restrict__ Ws) {
    using Cfg = GWCfg<NS>;
    constexpr int RS = Cfg::RS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA0 = smem;
    char* sA1 = smem + Cfg::A_BYTES;
    char* sB0 = smem + 2 * Cfg::A_BYTES;
    char* sB1 = sB0 + Cfg::B_BYTES;

    const int L = xcd_remap(blockIdx.x, p.nblk);
    const int tn = L % p.tiles_n, tm = L / p.tiles_n;
    const int m0 = tm * GB_BM, n0 = tn * GW_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;
    const int lrow = tid >> 2, lseg = tid & 3;
    const int nk = (p.K + GB_BK - 1) / GB_BK;
    f32x4 ra[2];             // X rows of the next K tile, in flight while the current tile is multiplied
    bool oka = true;         // zero fill is applied when the tile is written to LDS, not on the load
    const float xs = x_row_scale<NS>(p, m0);          // one range group per tile (ev2h_gemm_bf16 sends other shapes to the generic kernel)

    auto gload = [&](int kt) {
        const int k = kt * GB_BK + lseg * 8;
        int tap = 0, kc = k;
        if (p.taps == 3) { tap = k / p.Kc; kc = k - tap * p.Kc; }
        const int m = m0 + lrow;
        bool ok = (m < p.M) && (k < p.K);
        long src = m;
        if (p.taps == 3) {
            const int pos = m % p.rows_per_seq + tap - 1;
            ok = ok && (pos >= 0) && (pos < p.rows_per_seq);
            src = (long)m + tap - 1;
        }
        const float* g = p.X + (ok ? src : 0) * p.ldx + (ok ? kc : 0);
        // Issued from inline asm: once an LDS-DMA is in flight hipcc waits vmcnt(0) at (and before) every ordinary
        // load it tracks, which would make the DMA synchronous.  These loads are waited for by hand (wait_all).
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                     : "=&v"(ra[0]), "=&v"(ra[1]) : "v"(g) : "memory");
        oka = ok;
    };
    auto wait_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]) : : "memory"); };
    auto swrite = [&](char* dst) {
        f32x4 r[2] = {ra[0], ra[1]};
        if (!oka) { r[0] = f32x4{0.f, 0.f, 0.f, 0.f}; r[1] = r[0]; }
        scale_rows<NS>(r[0], xs); scale_rows<NS>(r[1], xs);
        unsigned q[4][plane_count(NS)];
        split_planes<NS>(r[0][0], r[0][1], q[0]);
        split_planes<NS>(r[0][2], r[0][3], q[1]);
        split_planes<NS>(r[1][0], r[1][1], q[2]);
        split_planes<NS>(r[1][2], r[1][3], q[3]);
#pragma unroll
        for (int s = 0; s < plane_count(NS); ++s) {
            u32x4 v = {q[0][s], q[1][s], q[2][s], q[3][s]};
            *reinterpret_cast<u32x4*>(dst + lrow * RS + s * 64 + lseg * 16) = v;
        }
    };
    auto dma_b = [&](int kt, char* dst) {
        const char* src = Ws + ((size_t)tn * nk + kt) * Cfg::B_BYTES;
        for (int off = wave * 1024; off < Cfg::B_BYTES; off += 8 * 1024)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + off), 16, 0, 0);
    };
    static_assert(Cfg::B_BYTES % 1024 == 0, "B tile image must be a whole number of 1 KiB DMA pieces");

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    dma_b(0, sB0);
    gload(0);
    wait_all();
    swrite(sA0);
    __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const char* sA = (kt & 1) ? sA1 : sA0;
        const char* sB = (kt & 1) ? sB1 : sB0;
        const bool more = kt + 1 < nk;
        if (more) {                     // next tile: W by LDS-DMA, X rows into registers; both land under the MFMAs
            dma_b(kt + 1, (kt & 1) ? sB0 : sB1);
            gload(kt + 1);
        }
        const char* pa = sA + (wm * 64 + l31) * RS + half * 16;
        const char* pb = sB + (wn * 64 + l31) * RS + half * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            u32x4 a[2][plane_count(NS)], b[2][plane_count(NS)];
#pragma unroll
            for (int s = 0; s < plane_count(NS); ++s) {
                a[0][s] = *reinterpret_cast<const u32x4*>(pa + s * 64 + m * 32);
                a[1][s] = *reinterpret_cast<const u32x4*>(pa + 32 * RS + s * 64 + m * 32);
                b[0][s] = *reinterpret_cast<const u32x4*>(pb + s * 64 + m * 32);
                b[1][s] = *reinterpret_cast<const u32x4*>(pb + 32 * RS + s * 64 + m * 32);
            }
            // plane-product outer, accumulator inner: consecutive MFMAs never depend on each other
#pragma unroll
            for (int q = 0; q < Planes<NS>::NPROD; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma_planes<NS>(a[i][Planes<NS>::A[q]], b[j][Planes<NS>::B[q]], acc[i][j]);
        }
        if (more) {
            wait_all();                 // X rows (and, being older, the DMA pieces of this wave) have landed
            swrite((kt & 1) ? sA0 : sA1);
        }
        __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    }

    gemm_epilogue<NS, 2, 2, false>(p, acc, m0, wm * 64, n0 + wn * 64, wm, wn * 64, GW_BN, reinterpret_cast<float*>(smem), tid);
}

template <int NS>
int launch_gw(const GemmBP& p, const char* Ws, hipStream_t st) {
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_wide_kernel<NS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GWCfg<NS>::LDS_BYTES)););
    gemm_nt_bf16_wide_kernel<NS><<<p.nblk, GB_THREADS, GWCfg<NS>::LDS_BYTES, st>>>(p, Ws);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

// ---------------------------------------------------------------------------------------- q1 never written
// Epilogue of the first query convolution when only the attention consumes it (TEHNet.py:150-166, 13-27; attention.hip): the second
// convolution is folded behind the attention's sum over the points, so q1 = BN(ReLU(conv(l0))) enters only through
//     Z[c][t][i] = sum_n key[c][n] q1[i][n + t - 1]            (key = segmentation logits, zero outside the window).
// A tile holds 128 rows n of 128 columns i in its accumulators -- in D layout exactly the B operand of a
// [12 (class, tap) rows] x [k = 128 tile rows] x [128 columns] product (a lane's registers 8 kb .. 8 kb + 7 are the 8 k slots of
// k-block kb; the key rows are gathered in that slot order) -- so the tile's contribution to Z is 8 more MFMAs per plane product
// and wave (+4 % on the tile's own), and the 1.07 GB of q1 per 256 windows are neither written nor read again.  The tile's
// 12 x 128 partial goes to zpart[(window, tile)][12][512]; attn_simfold_kernel adds the tiles in order (fixed order, no atomics:
// reproducible and batch-independent, like the two-pass form).  F16X2: q1 and the keys are scaled by powers of two chosen from the
// wave's / the tile's own maxima (exact), the partial is multiplied back.
template <int NS>
__device__ __forceinline__ void zsum_epilogue(const GemmBP& p, f32x16 (&acc)[2][2], int m0, int wm, int wn, int n0, char* smem, int tid) {
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    constexpr int KLD = 136;
    float* keyL = reinterpret_cast<float*>(smem);             // [4][KLD]: key[c][m0 - 1 + i], i = 0..129
    float* red = keyL + 4 * KLD;                               // [12][128]: the partial of the lower 64 rows' waves
    __syncthreads();                                           // the operand tiles are dead
    const int pos0 = m0 % p.rows_per_seq;
    for (int i = tid; i < GB_BM + 2; i += GO_THREADS_) {
        const int pos = pos0 - 1 + i;
        const long row = (long)m0 - 1 + i;
        float4 k = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pos >= 0 && pos < p.rows_per_seq && row < p.M) k = p.zs_key[row];
        keyL[i] = k.x; keyL[KLD + i] = k.y; keyL[2 * KLD + i] = k.z; keyL[3 * KLD + i] = k.w;
    }
    // q1 values of this wave: bias, ReLU, BN affine -- the store epilogue's arithmetic
    const float* bias = p.bias;
    float bj[2], sj[2], tj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const bool okc = col < p.N;
        bj[j] = (bias && okc) ? bias[col] : 0.f;
        sj[j] = (p.post_scale && okc) ? p.post_scale[col] : 1.f;
        tj[j] = (p.post_shift && okc) ? p.post_shift[col] : 0.f;
    }
    float cx = p.w_unscale;
    if (planes_f16(NS) && (p.x_amax || p.x_scale)) cx = p.w_unscale * pow2_inverse(x_row_scale<NS>(p, m0));
    float amf = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + mfma_row(r, half);
                float v = acc[i][j][r] * cx + bj[j];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.post_scale) v = __fmaf_rn(v, sj[j], tj[j]);
                if (row >= p.M) v = 0.f;
                acc[i][j][r] = v;
                amf = fmaxf(amf, fabsf(v));
            }
    __syncthreads();                                           // the key tile is complete
    float sq = 1.f, sk = 1.f;
    if constexpr (planes_f16(NS)) {
        sq = f16x2_scale(wave_max_u32_dpp(__float_as_uint(amf)));
        float km = 0.f;
        for (int i = lane; i < 4 * KLD; i += 64) km = fmaxf(km, (i % KLD) < GB_BM + 2 ? fabsf(keyL[i]) : 0.f);
        sk = f16x2_scale(wave_max_u32_dpp(__float_as_uint(km)));
    }
    // A operand: lane (m = l31 < 12: class m / 3, tap m % 3; the other lanes repeat m = 0 and their D rows are not used), half h:
    // k slot e of k-block (i, kb) is the tile row wm * 64 + 32 i + mfma_row(8 kb + e, h), paired with key row n - t + 1
    const int m = l31 < 12 ? l31 : 0, kc = m / 3, kt = m - 3 * kc;
    const float* kr = keyL + kc * KLD + 2 - kt + wm * 64;
    f32x16 z[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            u32x4 ap[plane_count(NS)];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float a0 = kr[32 * i + mfma_row(8 * kb + 2 * w, half)] * sk, a1 = kr[32 * i + mfma_row(8 * kb + 2 * w + 1, half)] * sk;
                unsigned o[plane_count(NS)];
                split_planes<NS>(a0, a1, o);
#pragma unroll
                for (int s_ = 0; s_ < plane_count(NS); ++s_) ap[s_][w] = o[s_];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x4 bq[plane_count(NS)];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    unsigned o[plane_count(NS)];
                    split_planes<NS>(acc[i][j][8 * kb + 2 * w] * sq, acc[i][j][8 * kb + 2 * w + 1] * sq, o);
#pragma unroll
                    for (int s_ = 0; s_ < plane_count(NS); ++s_) bq[s_][w] = o[s_];
                }
#pragma unroll
                for (int q = 0; q < Planes<NS>::NPROD; ++q) z[j] = mfma_planes<NS>(ap[Planes<NS>::A[q]], bq[Planes<NS>::B[q]], z[j]);
            }
        }
    // D rows: register r of half h is (class, tap) row mfma_row(r, h): rows 0..3 and 8..11 in the lower half-wave, 4..7 in the upper
    const float iq = pow2_inverse(sq), ik = pow2_inverse(sk);      // applied one after the other: their product may leave the fp32 range when the true sum does not
    const int ch = pos0 / GB_BM, nch = p.rows_per_seq / GB_BM;
    const long bw = m0 / p.rows_per_seq;
    float* zp = p.zs_out + ((size_t)bw * nch + ch) * 12 * p.N;
    if (wm == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int mr = mfma_row(r, half);
                if (mr < 12) red[mr * GB_BN_ + wn * 64 + j * 32 + l31] = (z[j][r] * iq) * ik;
            }
    }
    __syncthreads();
    if (wm == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int mr = mfma_row(r, half);
                const int col = n0 + wn * 64 + j * 32 + l31;
                if (mr < 12 && col < p.N) zp[(size_t)mr * p.N + col] = (z[j][r] * iq) * ik + red[mr * GB_BN_ + wn * 64 + j * 32 + l31];
            }
    }
}

// ---------------------------------------------------------------------------------------- occupancy variant
// 128 x 128 x 32 tiles, 4 waves (2 x 2, each 64 x 64), ONE LDS buffer (53 KB at NS = 3) so that three
// independent workgroups share a CU: while one sits at its barrier or stages the next K tile, the other two
// keep the matrix pipe busy (the barrier/refill bubble of a lone 8-wave workgroup costs 12-20 % here, see
// tools/ubench/mfma_lds.hip).  W arrives as 128-row plane images by LDS-DMA, X is split on the fly.
constexpr int GO_BN = 128, GO_THREADS = 256;

// TAP3 (Conv1d k=3 along the rows, Kc % 32 == 0, rows_per_seq % 128 == 0): the three taps of one 32-wide channel chunk
// are the same 130 rows of X shifted by one, so the chunk is loaded and split ONCE (128 rows + a one-row halo each side,
// zero at the window ends) and the three W tiles of that chunk are multiplied against row-shifted views of it.
template <int NS, bool TAP3>
struct GOCfg {
    static constexpr int RS = plane_count(NS) * 64 + 16;
    static constexpr int A_ROWS = GB_BM + (TAP3 ? 2 : 0);
    static constexpr int A_BYTES = A_ROWS * RS;
    static constexpr int B_BYTES = GO_BN * RS;
    static constexpr int LDS_BYTES = A_BYTES + B_BYTES;
};

// PIPE (small grids only, not TAP3): two LDS buffers; the X rows and the W tile of step k + 1 are requested BEFORE the MFMAs of step
// k and written behind them.  With three workgroups per CU the other two hide a workgroup's load -> split -> barrier chain; a launch
// of 2..8 workgroups (one window at a time: every M = 128 B layer) has nobody to hide it, and its time is (K / 32) x that chain.
// Same tiles, same MFMA order: bit-identical to the single-buffer kernel, so the choice (by launch size) never shows in a result.
// ZS (TAP3 only): the epilogue forms the attention's key-weighted sums instead of storing the tile (zsum_epilogue) -- its own
// instantiation, not a run-time choice inside one body (see sab_split_forms in sa_mlp_bf16.hip for what one body with two forms costs)
template <int NS, bool TAP3, bool PIPE = false, bool ZS = false>
__global__ __launch_bounds__(GO_THREADS, 3) void gemm_nt_bf16_occ_kernel(GemmBP p, const char* __restrict__ Ws) {
    static_assert(!ZS || TAP3, "the q1-free epilogue belongs to the tap kernel");
    using Cfg = GOCfg<NS, TAP3>;
    constexpr int RS = Cfg::RS;
    static_assert(!(PIPE && TAP3), "the pipelined variant is for the plain K loop");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + Cfg::A_BYTES;

    const int L = xcd_remap(blockIdx.x, p.nblk);
    const int tn = L % p.tiles_n, tm = L / p.tiles_n;
    const int m0 = tm * GB_BM, n0 = tn * GO_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int lrow = tid >> 1, lseg = tid & 1;            // loader: row, 16-wide k half
    const int nk = (p.K + GB_BK - 1) / GB_BK;
    f32x4 ra[4], rh[4];
    bool oka = true, okh = true, fulla = true;      // fulla: all 16 floats of the segment lie inside the row (k + 16 <= K)
    // one range group per tile (ev2h_gemm_bf16 sends other shapes to the generic kernel); the TAP3 halo rows m0 - 1 and
    // m0 + 128 lie in the tile's window whenever they are used, so they take the tile's scale too
    const float xs = x_row_scale<NS>(p, m0);
    const float xsh = xs;

    // 16 consecutive floats of one X row.  K is a multiple of 8, not of 16: when only the first 8 lie inside the row (`full`
    // false) the second half is read from the first half's address -- never past the row -- and zeroed when the tile is
    // written (swrite_row): the floats behind a row's end belong to the NEXT row, and although their weights are zero a
    // neighbour window's value times this window's power-of-two scale can overflow fp16 (inf * 0 = NaN).
    auto load16 = [&](const float* g, bool full, f32x4 (&r)[4]) {
        const float* g2 = full ? g + 8 : g;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                     "global_load_dwordx4 %2, %5, off\n\tglobal_load_dwordx4 %3, %5, off offset:16"
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(g), "v"(g2) : "memory");
    };
    auto gload = [&](int kt, f32x4 (&dst)[4]) {
        const int k = kt * GB_BK + lseg * 16;
        int tap = 0, kc = k;
        if (p.taps == 3) { tap = k / p.Kc; kc = k - tap * p.Kc; }
        const int m = m0 + lrow;
        bool ok = (m < p.M) && (k < p.K);
        long src = m;
        if (p.taps == 3) {
            const int pos = m % p.rows_per_seq + tap - 1;
            ok = ok && (pos >= 0) && (pos < p.rows_per_seq);
            src = (long)m + tap - 1;
        }
        fulla = (k + 16 <= p.K);
        load16(p.X + (ok ? src : 0) * p.ldx + (ok ? kc : 0), fulla, dst);
        oka = ok;
    };
    // TAP3: chunk kc of the centre rows (LDS rows 1..128) and, threads 0..3, of the two halo rows (LDS rows 0 and 129)
    // x_bf16 (NS = 1): 16 bf16 values = 32 bytes = the first two registers of ra / rh
    auto load16h = [&](const unsigned short* g, f32x4 (&r)[4]) {
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(r[0]), "=&v"(r[1]) : "v"(g) : "memory");
    };
    auto gload3 = [&](int kc) {
        const int k = kc * GB_BK + lseg * 16;
        const int m = m0 + lrow;
        oka = m < p.M;
        const bool xh = (NS == 1 || NS == 4) && p.x_bf16;
        if (xh) load16h(reinterpret_cast<const unsigned short*>(p.X) + (oka ? (long)m : 0) * p.ldx + k, ra);
        else load16(p.X + (oka ? (long)m : 0) * p.ldx + k, true, ra);            // TAP3: Kc % 32 == 0
        if (tid < 4) {
            const long mh = (tid >> 1) ? (long)m0 + GB_BM : (long)m0 - 1;
            okh = (tid >> 1) ? ((m0 + GB_BM) % p.rows_per_seq != 0 && mh < p.M) : (m0 % p.rows_per_seq != 0);
            if (xh) load16h(reinterpret_cast<const unsigned short*>(p.X) + (okh ? mh : 0) * p.ldx + k, rh);
            else load16(p.X + (okh ? mh : 0) * p.ldx + k, true, rh);
        }
    };
    auto wait_all = [&]() {
        if constexpr (TAP3)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(rh[0]), "+v"(rh[1]), "+v"(rh[2]), "+v"(rh[3]) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]) : : "memory");
    };
    auto swrite_row = [&](const f32x4 (&r)[4], bool ok, int ldsrow, float sc) {
        if constexpr ((NS == 1 || NS == 4) && TAP3) {
            if (p.x_bf16) {               // the loaded 32 bytes are the plane (F16: stored with the group's power of two): two 16-byte pieces of this row's k half
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    u32x4 v = __builtin_bit_cast(u32x4, r[hh]);
                    if (!ok) v = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4*>(sA + ldsrow * RS + lseg * 32 + hh * 16) = v;
                }
                return;
            }
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x4 r0 = r[2 * hh], r1 = r[2 * hh + 1];
            if (!ok || (hh == 1 && !fulla)) { r0 = f32x4{0.f, 0.f, 0.f, 0.f}; r1 = r0; }
            scale_rows<NS>(r0, sc); scale_rows<NS>(r1, sc);
            unsigned q[4][plane_count(NS)];
            split_planes<NS>(r0[0], r0[1], q[0]);
            split_planes<NS>(r0[2], r0[3], q[1]);
            split_planes<NS>(r1[0], r1[1], q[2]);
            split_planes<NS>(r1[2], r1[3], q[3]);
#pragma unroll
            for (int s = 0; s < plane_count(NS); ++s) {
                u32x4 v = {q[0][s], q[1][s], q[2][s], q[3][s]};
                *reinterpret_cast<u32x4*>(sA + ldsrow * RS + s * 64 + lseg * 32 + hh * 16) = v;
            }
        }
    };
    auto dma_b = [&](int kt) {
        const char* src = Ws + ((size_t)tn * nk + kt) * Cfg::B_BYTES;
        for (int off = wave * 1024; off < Cfg::B_BYTES; off += 4 * 1024)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + lane * 16),
                                             (__attribute__((address_space(3))) void*)(sB + off), 16, 0, 0);
    };
    static_assert(Cfg::B_BYTES % 1024 == 0, "B tile image must be a whole number of 1 KiB DMA pieces");

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mma_tile = [&](int arow) {
        const char* pa = sA + (wm * 64 + l31 + arow) * RS + half * 16;
        const char* pb = sB + (wn * 64 + l31) * RS + half * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            u32x4 a[2][plane_count(NS)], b[2][plane_count(NS)];
#pragma unroll
            for (int s = 0; s < plane_count(NS); ++s) {
                a[0][s] = *reinterpret_cast<const u32x4*>(pa + s * 64 + m * 32);
                a[1][s] = *reinterpret_cast<const u32x4*>(pa + 32 * RS + s * 64 + m * 32);
                b[0][s] = *reinterpret_cast<const u32x4*>(pb + s * 64 + m * 32);
                b[1][s] = *reinterpret_cast<const u32x4*>(pb + 32 * RS + s * 64 + m * 32);
            }
            // plane-product outer, accumulator inner: consecutive MFMAs never depend on each other
#pragma unroll
            for (int q = 0; q < Planes<NS>::NPROD; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma_planes<NS>(a[i][Planes<NS>::A[q]], b[j][Planes<NS>::B[q]], acc[i][j]);
        }
        __builtin_amdgcn_s_waitcnt(0x0070);              // all fragment reads returned before a tile is overwritten
    };
    if constexpr (TAP3) {
        const int nkc = p.Kc / GB_BK;
        for (int kc = 0; kc < nkc; ++kc) {
            if (kc) __builtin_amdgcn_s_barrier();        // everyone finished reading the previous chunk
            dma_b(kc);
            gload3(kc);
            wait_all();
            swrite_row(ra, oka, lrow + 1, xs);
            if (tid < 4) swrite_row(rh, okh, (tid >> 1) ? GB_BM + 1 : 0, xsh);
            __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
#pragma unroll 1
            for (int tap = 0; tap < 3; ++tap) {
                if (tap) {
                    __builtin_amdgcn_s_barrier();        // sB is free
                    dma_b(tap * nkc + kc);
                    __builtin_amdgcn_s_waitcnt(0x0070);
                    __builtin_amdgcn_s_barrier();
                }
                mma_tile(tap);                           // output row j of tap t reads X row j + t - 1 = LDS row j + t
            }
        }
    } else if constexpr (PIPE) {
        // KT K-tiles (32 columns each) per step and buffer.  KT = 2 (to amortise the ~1.1 us issue-to-landing time of an LDS-DMA piece,
        // MI355X_MICROARCH.md) measured the same as KT = 1 (B = 1 forward 1.166 vs 1.157 ms, K = 520 layer 32 us either way): with four
        // waves on one CU a 128 x 128 x 32 tile step is bound by its own split + 24 MFMAs per wave (~1.2 us), not by the transport.
        // KT = 1 keeps the footprint at 74 KB.  The tiles are multiplied in K order as in the single-buffer kernel.
        constexpr int KT = 1;
        char* const base = smem;
        auto use = [&](int bufi, int j) { sA = base + (bufi * KT + j) * Cfg::LDS_BYTES; sB = sA + Cfg::A_BYTES; };
        f32x4 rq[KT][4];
        bool okq[KT], fullq[KT];
        auto stage = [&](int kt0, int bufi) {            // request the tiles kt0 .. kt0 + KT - 1 (W by LDS-DMA, X into registers)
#pragma unroll
            for (int j = 0; j < KT; ++j)
                if (kt0 + j < nk) {
                    use(bufi, j);
                    dma_b(kt0 + j);
                    gload(kt0 + j, rq[j]);                // (straight into this tile's registers: they are valid only after commit's wait)
                    okq[j] = oka; fullq[j] = fulla;
                }
        };
        auto commit = [&](int kt0, int bufi) {           // the X rows have arrived: split them into the buffer
            if constexpr (KT == 2)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rq[0][0]), "+v"(rq[0][1]), "+v"(rq[0][2]), "+v"(rq[0][3]), "+v"(rq[1][0]), "+v"(rq[1][1]), "+v"(rq[1][2]), "+v"(rq[1][3]) : : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rq[0][0]), "+v"(rq[0][1]), "+v"(rq[0][2]), "+v"(rq[0][3]) : : "memory");
#pragma unroll
            for (int j = 0; j < KT; ++j)
                if (kt0 + j < nk) {
                    use(bufi, j);
                    fulla = fullq[j];
                    swrite_row(rq[j], okq[j], lrow, xs);
                }
            __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
        };
        stage(0, 0);
        commit(0, 0);
        int bufi = 0;
        for (int kt = 0; kt < nk; kt += KT, bufi ^= 1) {
            __builtin_amdgcn_s_barrier();                // buffer `bufi` is complete; nobody reads the other one any more
            const bool more = kt + KT < nk;
            if (more) stage(kt + KT, bufi ^ 1);
#pragma unroll
            for (int j = 0; j < KT; ++j)
                if (kt + j < nk) { use(bufi, j); mma_tile(0); }
            if (more) commit(kt + KT, bufi ^ 1);
        }
        __builtin_amdgcn_s_barrier();                    // the epilogue reuses LDS (row-max reduction)
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            if (kt) __builtin_amdgcn_s_barrier();        // everyone finished reading the previous tile
            dma_b(kt);
            gload(kt, ra);
            wait_all();
            swrite_row(ra, oka, lrow, xs);
            __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            mma_tile(0);
        }
    }

    if constexpr (ZS) {
        zsum_epilogue<NS>(p, acc, m0, wm, wn, n0, smem, tid);
        return;
    }
#ifdef EV2H_GEMM_ONE_BODY            // build A/B: both epilogues in the tap kernel's one body, chosen at run time (the form until round 5)
    if constexpr (TAP3) {
        if (p.zs_out) { zsum_epilogue<NS>(p, acc, m0, wm, wn, n0, smem, tid); return; }
    }
#endif
    gemm_epilogue<NS, 2, 2, false>(p, acc, m0, wm * 64, n0 + wn * 64, wm, wn * 64, GO_BN, reinterpret_cast<float*>(smem), tid);
}

template <int NS, bool TAP3, bool ZS = false>
int launch_go_t(const GemmBP& p, const char* Ws, hipStream_t st) {
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_occ_kernel<NS, TAP3, false, ZS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GOCfg<NS, TAP3>::LDS_BYTES)););
    gemm_nt_bf16_occ_kernel<NS, TAP3, false, ZS><<<p.nblk, GO_THREADS, GOCfg<NS, TAP3>::LDS_BYTES, st>>>(p, Ws);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

// ---------------------------------------------------------------------------------------- small-grid variant
// 64 x 64 x 32 tiles, 4 waves (2 x 2, each ONE 32 x 32 accumulator), two LDS buffers, for launches whose 128 x 128 tiling gives fewer
// than 65 workgroups (one to a few windows at a time: the M = 128 B layers).  There a K step of the big tile -- the split of 128 rows
// plus 24 MFMAs per wave, ~1.2 us with nobody else on the CU -- is the latency of the whole layer (17 steps for K = 520: 30 us, eleven
// such layers on the critical path of a one-window forward); a quarter-size tile puts four times the workgroups on the idle chip.
// Every output element accumulates the same products in the same order as in the big-tile kernels (k tiles ascending, the two
// 16-column k blocks, the plane products): bit-identical, so the choice (by launch size) never shows in a result.
template <int NS>
struct GSCfg {
    static constexpr int RS = plane_count(NS) * 64 + 16;
    static constexpr int A_BYTES = 64 * RS, B_BYTES = 64 * RS, BUF = A_BYTES + B_BYTES, LDS_BYTES = 2 * BUF;
};

template <int NS>
__global__ __launch_bounds__(GO_THREADS, 3) void gemm_nt_bf16_small_kernel(GemmBP p, const char* __restrict__ Ws) {
    using Cfg = GSCfg<NS>;
    constexpr int RS = Cfg::RS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = xcd_remap(blockIdx.x, p.nblk);
    const int tn = L % p.tiles_n, tm = L / p.tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int lrow = tid >> 2, lseg = tid & 3;            // loader: row, 8-wide k quarter
    const int nk = (p.K + GB_BK - 1) / GB_BK;
    const float xs = x_row_scale<NS>(p, m0);
    f32x4 r0, r1;
    bool ok = true;
    auto gload = [&](int kt) {
        const int k = kt * GB_BK + lseg * 8, m = m0 + lrow;
        ok = (m < p.M) && (k < p.K);                       // K % 8 == 0: a segment lies inside the row or behind it
        const float* g = p.X + (ok ? (long)m : 0) * p.ldx + (ok ? k : 0);
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(r0), "=&v"(r1) : "v"(g) : "memory");
    };
    auto commit = [&](char* A) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1) : : "memory");
        f32x4 a = r0, b = r1;
        if (!ok) { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
        scale_rows<NS>(a, xs); scale_rows<NS>(b, xs);
        unsigned q[4][plane_count(NS)];
        split_planes<NS>(a[0], a[1], q[0]);
        split_planes<NS>(a[2], a[3], q[1]);
        split_planes<NS>(b[0], b[1], q[2]);
        split_planes<NS>(b[2], b[3], q[3]);
#pragma unroll
        for (int s_ = 0; s_ < plane_count(NS); ++s_) {
            u32x4 v = {q[0][s_], q[1][s_], q[2][s_], q[3][s_]};
            *reinterpret_cast<u32x4*>(A + lrow * RS + s_ * 64 + lseg * 16) = v;
        }
        __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0)
    };
    auto dma_b = [&](int kt, char* B) {                    // this tile's 64 rows of the 128-row image tile
        const char* src = Ws + ((size_t)(tn >> 1) * nk + kt) * (size_t)(GO_BN * RS) + (size_t)(tn & 1) * Cfg::B_BYTES;
        for (int off = wave * 1024; off < Cfg::B_BYTES; off += 4 * 1024)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + lane * 16),
                                             (__attribute__((address_space(3))) void*)(B + off), 16, 0, 0);
    };
    static_assert(Cfg::B_BYTES % 1024 == 0, "half an image tile must be a whole number of 1 KiB DMA pieces");
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    auto mma = [&](const char* A, const char* B) {
        const char* pa = A + (wm * 32 + l31) * RS + half * 16;
        const char* pb = B + (wn * 32 + l31) * RS + half * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            u32x4 a[plane_count(NS)], b[plane_count(NS)];
#pragma unroll
            for (int s_ = 0; s_ < plane_count(NS); ++s_) {
                a[s_] = *reinterpret_cast<const u32x4*>(pa + s_ * 64 + m * 32);
                b[s_] = *reinterpret_cast<const u32x4*>(pb + s_ * 64 + m * 32);
            }
#pragma unroll
            for (int q = 0; q < Planes<NS>::NPROD; ++q) acc[0][0] = mfma_planes<NS>(a[Planes<NS>::A[q]], b[Planes<NS>::B[q]], acc[0][0]);
        }
        __builtin_amdgcn_s_waitcnt(0x0070);              // all fragment reads returned before a tile is overwritten
    };
    dma_b(0, smem + Cfg::A_BYTES);
    gload(0);
    commit(smem);
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();                    // buffer kt & 1 is complete; nobody reads the other one any more
        const bool more = kt + 1 < nk;
        char* cur = smem + (kt & 1) * Cfg::BUF;
        char* nxt = smem + ((kt + 1) & 1) * Cfg::BUF;
        if (more) { dma_b(kt + 1, nxt + Cfg::A_BYTES); gload(kt + 1); }
        mma(cur, cur + Cfg::A_BYTES);
        if (more) commit(nxt);
    }
    __builtin_amdgcn_s_barrier();
    gemm_epilogue<NS, 1, 1, false>(p, acc, m0, wm * 32, n0 + wn * 32, wm, wn * 32, 64, reinterpret_cast<float*>(smem), tid);
}

template <int NS>
int launch_go_small(GemmBP p, const char* Ws, hipStream_t st) {
    p.tiles_n = ceil_div(p.N, 64);
    p.nblk = ceil_div(p.M, 64) * p.tiles_n;
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_small_kernel<NS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GSCfg<NS>::LDS_BYTES)););
    gemm_nt_bf16_small_kernel<NS><<<p.nblk, GO_THREADS, GSCfg<NS>::LDS_BYTES, st>>>(p, Ws);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

template <int NS>
constexpr int go_pipe_lds() { return 2 * GOCfg<NS, false>::LDS_BYTES; }

template <int NS>
int launch_go_pipe(const GemmBP& p, const char* Ws, hipStream_t st) {
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_occ_kernel<NS, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, go_pipe_lds<NS>())););
    gemm_nt_bf16_occ_kernel<NS, false, true><<<p.nblk, GO_THREADS, go_pipe_lds<NS>(), st>>>(p, Ws);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

template <int NS>
int launch_go(const GemmBP& p, const char* Ws, hipStream_t st) {
    // (the tap-reuse kernel for the shapes that tile; every other k = 3 shape -- N % 128 != 0 -- takes the plain K loop below, which the
    //  odd-N end-to-end cases exercise: the EV2H_GEMM_NO_TAP3 switch of rounds 2-5 is retired)
    if (p.taps == 3 && p.Kc % GB_BK == 0 && p.rows_per_seq % GB_BM == 0 && p.M % p.rows_per_seq == 0)
        return launch_go_t<NS, true>(p, Ws, st);
    // (same sums in the same order in all three tilings: test_gpu_ops.py::test_gemm_small_grids_bit_identical compares them through M)
    if (p.nblk <= 64 && p.rowmax_rows == 0 && p.taps == 1) return launch_go_small<NS>(p, Ws, st);   // quarter-size tiles
    if (p.nblk <= 128) return launch_go_pipe<NS>(p, Ws, st);      // fewer workgroups than CUs: hide the K-step chain inside the workgroup
    return launch_go_t<NS, false>(p, Ws, st);
}

}  // namespace

// geometry of the W plane images of the fast kernels (ev2h_tile_geometry): bytes per LDS row, K columns per tile
int ev2h_gemm_tile_geometry(int ns, int out[2]) {
    out[0] = ns == 1 ? GBCfg<1>::RS : ns == 2 ? GBCfg<2>::RS : ns == 3 ? GBCfg<3>::RS : GBCfg<4>::RS;
    out[1] = GB_BK;
    return EV2H_OK;
}

// The first query convolution with the attention's key-weighted sums as its output (zsum_epilogue): d describes the k = 3 GEMM as for
// ev2h_gemm (d->Y unused), key_pm = logits point-major [M][4], zpart [M / 128][12][N].  Internal (forward.hip).
// ev2h_gemm_bf16_zsum_supported: the shapes the tap kernel takes -- the caller tests them FIRST and runs the two-pass form otherwise;
// an error of the launch itself is then an error, not a silent change of schedule.
bool ev2h_gemm_bf16_zsum_supported(const ev2h_gemm_desc* d) {
    if (!(d->taps == 3 && d->Ws && d->ws_tile_rows == 128 && (d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_BF16 || d->precision == EV2H_PREC_BF16X3 || d->precision == EV2H_PREC_F16) &&
          d->K % GB_BK == 0 && d->rows_per_seq > 0 && d->rows_per_seq % GB_BM == 0 && d->M % d->rows_per_seq == 0 && d->N % GO_BN == 0))
        return false;
    if ((d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_F16) && d->x_amax) {
        const int xg = d->x_group_rows > 0 ? d->x_group_rows : 1;
        if (xg % GB_BM != 0 || xg % d->rows_per_seq != 0) return false;
    }
    return true;
}

int ev2h_gemm_bf16_zsum(const ev2h_gemm_desc* d, const float* key_pm, float* zpart, int x_bf16, ev2h_stream_t stream, const float* x_scale) {
    EV2H_CHECK_ARG(d && key_pm && zpart && ev2h_gemm_bf16_zsum_supported(d));
    EV2H_CHECK_ARG(!x_bf16 || d->precision == EV2H_PREC_BF16 || (d->precision == EV2H_PREC_F16 && x_scale && d->x_group_rows > 0));
    GemmBP p{};
    p.X = d->X; p.ldx = d->ldx; p.W = d->W; p.ldw = d->ldw; p.Y = nullptr; p.ldy = 0;
    p.M = d->M; p.N = d->N; p.taps = 3; p.Kc = d->K; p.K = d->K * 3;
    p.rows_per_seq = d->rows_per_seq;
    p.bias = d->bias; p.relu = d->relu; p.post_scale = d->post_scale; p.post_shift = d->post_shift;
    p.w_unscale = d->w_unscale > 0.f ? d->w_unscale : 1.f;
    p.x_group_rows = p.y_group_rows = 1;
    if ((d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_F16) && d->x_amax) {
        p.x_amax = d->x_amax; p.x_amax2 = d->x_amax2; p.x_group_rows = d->x_group_rows > 0 ? d->x_group_rows : 1;
    }
    p.zs_key = reinterpret_cast<const float4*>(key_pm); p.zs_out = zpart;
    p.x_bf16 = x_bf16;
    if (x_bf16 && d->precision == EV2H_PREC_F16) { p.x_scale = x_scale; p.x_amax = p.x_amax2 = nullptr; p.x_group_rows = d->x_group_rows; }
    p.tiles_n = d->N / GO_BN;
    p.nblk = (d->M / GB_BM) * p.tiles_n;
#ifdef EV2H_GEMM_ONE_BODY
    constexpr bool ZS = false;
#else
    constexpr bool ZS = true;
#endif
    if (d->precision == EV2H_PREC_F16X2) return launch_go_t<2, true, ZS>(p, (const char*)d->Ws, (hipStream_t)stream);
    if (d->precision == EV2H_PREC_BF16) return launch_go_t<1, true, ZS>(p, (const char*)d->Ws, (hipStream_t)stream);
    if (d->precision == EV2H_PREC_BF16X3) return launch_go_t<3, true, ZS>(p, (const char*)d->Ws, (hipStream_t)stream);
    if (d->precision == EV2H_PREC_F16) return launch_go_t<4, true, ZS>(p, (const char*)d->Ws, (hipStream_t)stream);
    return EV2H_ERR_ARG;
}

// called by ev2h_gemm when d->precision != EV2H_PREC_F32 (arguments already validated there)
int ev2h_gemm_bf16(const ev2h_gemm_desc* d, ev2h_stream_t stream) {
    EV2H_CHECK_ARG((d->K % 8) == 0);
    if (d->taps == 3) EV2H_CHECK_ARG((d->K % 16) == 0);       // a 16-float loader segment never straddles two taps
    GemmBP p{};
    p.X = d->X; p.ldx = d->ldx; p.W = d->W; p.ldw = d->ldw; p.Y = d->Y; p.ldy = d->ldy;
    p.M = d->M; p.N = d->N; p.taps = d->taps; p.Kc = d->K; p.K = d->K * d->taps;
    p.rows_per_seq = d->rows_per_seq;
    p.bias = d->bias; p.bias_group_rows = d->bias_group_rows; p.ldbias = d->ldbias;
    p.relu = d->relu; p.post_scale = d->post_scale; p.post_shift = d->post_shift;
    p.rowmax_rows = d->rowmax_rows;
    p.w_unscale = d->w_unscale > 0.f ? d->w_unscale : 1.f;
    if (d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_F16) {
        p.x_amax = d->x_amax; p.x_amax2 = d->x_amax2; p.x_group_rows = d->x_group_rows > 0 ? d->x_group_rows : 1;
        p.y_amax = d->y_amax; p.y_group_rows = d->y_group_rows > 0 ? d->y_group_rows : 1;
        p.y_scale = d->y_scale; p.y_bound_w = d->y_bound_w; p.y_bound_b = d->y_bound_b;
        if (!p.x_amax) { p.x_amax2 = nullptr; p.y_scale = nullptr; }
        if (p.y_scale) EV2H_CHECK_ARG(p.x_group_rows == p.y_group_rows && d->rowmax_rows == 0);
        if (d->taps == 3 && p.x_amax) EV2H_CHECK_ARG(p.x_group_rows % d->rows_per_seq == 0);   // a sequence never straddles two scales
    } else {
        p.x_group_rows = p.y_group_rows = 1;
    }
    // range groups that do not tile by 128 rows (N % 128 != 0, or one row per group: the per-window head layers) take the
    // generic kernel with the per-row epilogue; everything on the hot path is tile aligned
    const bool general = (p.x_amax && p.x_group_rows % GB_BM != 0) ||
                         (p.y_amax && d->rowmax_rows == 0 && p.y_group_rows % GB_BM != 0);
    if (general) {
        p.tiles_n = ceil_div(d->N, GB_BN);
        p.nblk = ceil_div(d->M, GB_BM) * p.tiles_n;
        return d->precision == EV2H_PREC_F16 ? launch_gb<4, true>(p, (hipStream_t)stream) : launch_gb<2, true>(p, (hipStream_t)stream);
    }
    if (d->Ws && d->ws_tile_rows == 128) {   // 128-row plane images: three small workgroups per CU
        p.tiles_n = ceil_div(d->N, GO_BN);
        p.nblk = ceil_div(d->M, GB_BM) * p.tiles_n;
        if (d->precision == EV2H_PREC_BF16X3) return launch_go<3>(p, (const char*)d->Ws, (hipStream_t)stream);
        if (d->precision == EV2H_PREC_F16X2) return launch_go<2>(p, (const char*)d->Ws, (hipStream_t)stream);
        if (d->precision == EV2H_PREC_BF16) return launch_go<1>(p, (const char*)d->Ws, (hipStream_t)stream);
        if (d->precision == EV2H_PREC_F16) return launch_go<4>(p, (const char*)d->Ws, (hipStream_t)stream);
    }
    if (d->Ws) {   // host-packed plane images of W: wide tile, W streamed by LDS-DMA
        p.tiles_n = ceil_div(d->N, GW_BN);
        p.nblk = ceil_div(d->M, GB_BM) * p.tiles_n;
        if (d->precision == EV2H_PREC_BF16X3) return launch_gw<3>(p, (const char*)d->Ws, (hipStream_t)stream);
        if (d->precision == EV2H_PREC_F16X2) return launch_gw<2>(p, (const char*)d->Ws, (hipStream_t)stream);
        if (d->precision == EV2H_PREC_BF16) return launch_gw<1>(p, (const char*)d->Ws, (hipStream_t)stream);
        if (d->precision == EV2H_PREC_F16) return launch_gw<4>(p, (const char*)d->Ws, (hipStream_t)stream);
    }
    p.tiles_n = ceil_div(d->N, GB_BN);
    p.nblk = ceil_div(d->M, GB_BM) * p.tiles_n;
    if (d->precision == EV2H_PREC_BF16X3) return launch_gb<3, false>(p, (hipStream_t)stream);
    if (d->precision == EV2H_PREC_F16X2) return launch_gb<2, false>(p, (hipStream_t)stream);
    if (d->precision == EV2H_PREC_BF16) return launch_gb<1, false>(p, (hipStream_t)stream);
    if (d->precision == EV2H_PREC_F16) return launch_gb<4, false>(p, (hipStream_t)stream);
    ev2h_set_error("ev2h_gemm: unknown precision %d", d->precision);
    return EV2H_ERR_ARG;
}
