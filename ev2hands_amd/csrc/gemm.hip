// Point-wise dense layers of the Ev2Hands path as one fp32 MFMA "NT" GEMM for gfx950:
//     Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )
// X is point-major activations (rows = points of all windows, K contiguous), W is the PyTorch
// [out, in] weight (BN folded by the host).  Covers every Conv1d/Conv2d(1x1)/Linear of
// TEHNet.py:127-166 that is not inside a grouped set-abstraction MLP, the k=3 query convolutions
// (TEHNet.py:150-166, three shifted row taps, zero padded per window) and the group-all
// set-abstraction max (pointnet2_utils.py:195-200, row-max epilogue).
//
// 128x128 output tile per 256-thread workgroup, BK=32, 4 waves as 2x2, each wave 64x64 =
// 2x2 v_mfma_f32_32x32x2_f32 tiles (exact fp32, 64 accumulator VGPRs).  Operand tiles are staged
// in LDS with a (BK+4)-float row stride so the ds_read_b128 fragment reads are conflict free;
// the next K tile is prefetched into registers while the current one is multiplied; two LDS
// buffers -> one barrier per K tile.
#include <cstdlib>
#include "common.hpp"
#include "ev2hands_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;
constexpr int GEMM_LDS_BYTES = 2 * (BM + BN) * LDT * (int)sizeof(float);

struct GemmP {
    const float* X; int ldx;
    const float* W; int ldw;
    float* Y; int ldy;
    int M, N, K;           // K = taps * Kc
    const float* bias; int bias_group_rows; int ldbias;
    int relu;
    const float* post_scale; const float* post_shift;
    int taps; int Kc; int rows_per_seq;
    int rowmax_rows;
    int tiles_n; int nblk;
};

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    float* sA0 = smem;
    float* sB0 = smem + BM * LDT;
    float* sA1 = smem + (BM + BN) * LDT;
    float* sB1 = sA1 + BM * LDT;

    const int L = xcd_remap(blockIdx.x, p.nblk);
    const int tn = L % p.tiles_n, tm = L / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // loader mapping: 8 threads cover one 32-float row (float4 each), 32 rows per pass, 4 passes
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    float4 ra[4], rb[4];

    auto gload = [&](int kt) {
        const int k = kt * BK + lcol;
        int tap = 0, kc = k;
        if (p.taps == 3) { tap = k / p.Kc; kc = k - tap * p.Kc; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lrow + 32 * i;
            bool ok = (m < p.M) && (k < p.K);
            long src = m;
            if (p.taps == 3) {
                const int pos = m % p.rows_per_seq + tap - 1;
                ok = ok && (pos >= 0) && (pos < p.rows_per_seq);
                src = (long)m + tap - 1;
            }
            ra[i] = ok ? *reinterpret_cast<const float4*>(p.X + src * p.ldx + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + lrow + 32 * i;
            const bool ok = (n < p.N) && (k < p.K);
            rb[i] = ok ? *reinterpret_cast<const float4*>(p.W + (long)n * p.ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto swrite = [&](float* sA, float* sB) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(sA + (lrow + 32 * i) * LDT + lcol) = ra[i];
            *reinterpret_cast<float4*>(sB + (lrow + 32 * i) * LDT + lcol) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    swrite(sA0, sB0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const float* sA = (kt & 1) ? sA1 : sA0;
        const float* sB = (kt & 1) ? sB1 : sB0;
        if (kt + 1 < nk) gload(kt + 1);
        const float* pa = sA + (wm * 64 + l31) * LDT + half * 4;
        const float* pb = sB + (wn * 64 + l31) * LDT + half * 4;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const float4 a0 = *reinterpret_cast<const float4*>(pa + kb * 8);
            const float4 a1 = *reinterpret_cast<const float4*>(pa + 32 * LDT + kb * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(pb + kb * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(pb + 32 * LDT + kb * 8);
            const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(av[i][s], bv[j][s], acc[i][j]);
        }
        if (kt + 1 < nk) swrite((kt & 1) ? sA0 : sA1, (kt & 1) ? sB0 : sB1);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    const float* bias = p.bias;
    if (bias && p.bias_group_rows > 0) bias += (long)(m0 / p.bias_group_rows) * p.ldbias;
    float bj[2], sj[2], tj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const bool okc = col < p.N;
        bj[j] = (bias && okc) ? bias[col] : 0.f;
        sj[j] = (p.post_scale && okc) ? p.post_scale[col] : 1.f;
        tj[j] = (p.post_shift && okc) ? p.post_shift[col] : 0.f;
    }
    if (p.rowmax_rows == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + mfma_row(r, half);
                    float v = acc[i][j][r] + bj[j];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.post_scale) v = __fmaf_rn(v, sj[j], tj[j]);
                    if (row < p.M && col < p.N) p.Y[(long)row * p.ldy + col] = v;
                }
            }
    } else {
        // group-all max over the tile's 128 rows (one window per tile: rowmax_rows == BM)
        float* red = smem;   // operand tiles are dead after the last barrier of the K loop
        float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + mfma_row(r, half);
                    float v = acc[i][j][r] + bj[j];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.post_scale) v = __fmaf_rn(v, sj[j], tj[j]);
                    if (row < p.M) mx[j] = fmaxf(mx[j], v);
                }
#pragma unroll
        for (int j = 0; j < 2; ++j) mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], 32, 64));
        if (half == 0) {
            red[wm * BN + wn * 64 + l31] = mx[0];
            red[wm * BN + wn * 64 + 32 + l31] = mx[1];
        }
        __syncthreads();
        if (tid < BN) {
            const int col = n0 + tid;
            if (col < p.N) p.Y[(long)(m0 / p.rowmax_rows) * p.ldy + col] = fmaxf(red[tid], red[BN + tid]);
        }
    }
}


// ---------------------------------------------------------------------------------------- K = 8 layers
// The layer-1 feature tables of the set abstractions that read the raw cloud (enc.sa1, both MANO regressors:
// pointnet2_utils.py:248,253 with 4-5 feature channels padded to 8) are 8 -> 160 / 256 linear maps over B*N rows: 8 MACs per
// 4-byte output, i.e. bound by writing the table (0.5 GB at B = 256).  A K = 32 MFMA tile wastes three quarters of the matrix
// pipe on it and pays a 128-row tile prologue; here a lane owns four output columns (32 weights in registers), a wave walks
// 32 consecutive rows whose 8 inputs arrive by scalar loads, and every row is one 1 KiB coalesced store.  Arithmetic: the fp32
// fma chain in k order -- bit-identical to the f32 MFMA kernel -- in EVERY precision mode (the plane-split modes would only
// approximate it).  F16X2 range handling as in gemm_bf16.hip (storage scale from a bound, output record).
struct TableP {
    const float* X; int ldx;
    const float* W; int ldw;
    float* Y; int ldy;
    int M, N;
    const float* bias;
    int relu;
    const unsigned* x_amax; const unsigned* x_amax2; int x_group_rows;
    unsigned* y_amax; int y_group_rows;
    float* y_scale; float y_bound_w, y_bound_b;
};

constexpr int TB_ROWS_PER_WAVE = 32, TB_WAVES = 4;

__device__ __forceinline__ float tb_f16x2_scale(unsigned amax_bits) {      // planes.hpp: f16x2_scale
    const int E = (int)((amax_bits >> 23) & 0xffu);
    const int sb = (E == 255) ? 127 : min(268 - E, 200);
    return __uint_as_float((unsigned)sb << 23);
}

__global__ __launch_bounds__(256) void table_k8_kernel(TableP p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row_lo = ((long)blockIdx.x * TB_WAVES + wave) * TB_ROWS_PER_WAVE;
    if (row_lo >= p.M) return;
    const long row_hi = min((long)p.M, row_lo + TB_ROWS_PER_WAVE);
    for (int cb = 0; cb < p.N; cb += 256) {
        const int c0 = cb + 4 * lane;
        const bool okc = c0 < p.N;                 // N % 4 == 0
        float w[4][8], bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4* wr = reinterpret_cast<const float4*>(p.W + (long)(okc ? c0 + j : 0) * p.ldw);
            const float4 a = wr[0], b = wr[1];
            w[j][0] = a.x; w[j][1] = a.y; w[j][2] = a.z; w[j][3] = a.w; w[j][4] = b.x; w[j][5] = b.y; w[j][6] = b.z; w[j][7] = b.w;
            bj[j] = (p.bias && okc) ? p.bias[c0 + j] : 0.f;
        }
        long next = row_lo;                        // first row of the next range group
        long g = 0;
        float sy = 1.f, am = 0.f;
        auto flush = [&]() {                       // the finished group's maximum -> its record
            if (p.y_amax) {
                const unsigned m = wave_max_u32_dpp(__float_as_uint(am));
                if (lane == 0 && m) atomicMax(&p.y_amax[g], m);
            }
            am = 0.f;
        };
        for (long row = row_lo; row < row_hi; ++row) {
            if (row >= next && (p.y_scale || p.y_amax)) {      // wave-uniform
                if (row > row_lo) flush();
                const int gr = p.y_amax ? p.y_group_rows : p.x_group_rows;
                g = row / gr;
                next = (g + 1) * gr;
                if (p.y_scale) {
                    unsigned a = p.x_amax[g];
                    if (p.x_amax2) a = max(a, p.x_amax2[g]);
                    sy = tb_f16x2_scale(__float_as_uint(__fmaf_rn(p.y_bound_w, __uint_as_float(a), p.y_bound_b)));
                    if (lane == 0 && cb == 0 && row == g * gr) p.y_scale[g] = sy;
                }
            }
            const float4* xr = reinterpret_cast<const float4*>(p.X + row * p.ldx);       // wave-uniform address: scalar loads
            const float4 xa = xr[0], xb = xr[1];
            const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            float4 o;
            float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = __fmaf_rn(x[k], w[j][k], acc);
                float v = acc + bj[j];
                if (p.relu) v = fmaxf(v, 0.f);
                v *= sy;
                ov[j] = v;
                am = fmaxf(am, fabsf(v));
            }
            if (okc) *reinterpret_cast<float4*>(p.Y + row * p.ldy + c0) = o;
            else am = 0.f;
        }
        if (p.y_amax) flush();
    }
}

// tiny-N / odd-shape fallback is not needed: every other layer of the path goes through the tile above.

// logits [B*N][4] point-major -> class_logits [B,4,N] (TEHNet.py:188 output layout)
__global__ __launch_bounds__(256) void transpose_logits_kernel(const float4* __restrict__ pm, int N, float* __restrict__ cm, size_t cm_stride) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float4 v = pm[(size_t)b * N + n];
    float* o = cm + (size_t)b * cm_stride + n;
    o[0] = v.x; o[(size_t)N] = v.y; o[(size_t)2 * N] = v.z; o[(size_t)3 * N] = v.w;
}

// ---------------------------------------------------------------------------------------- one row per window
// The layers whose M is the number of WINDOWS (the broadcast half of fp3, the two Linear layers of each MANO head: M = B, K = 512 or
// 1024) leave a 128 x 128-tile kernel with 2-16 workgroups walking a 16-32 step K loop: 55-70 us each, on the critical path, and
// the same at B = 1.  Here the parallelism comes from K (see the kernel); the summation order is fixed and batch-independent, and
// the arithmetic is exact fp32 fma chains in every precision mode, so no operand planes and no range scaling are involved.
// Chosen by the caller (ev2h_gemm_desc.skinny), never by M: a window's result must not depend on the batch size.
struct SkinnyP {
    const float* X; int ldx;
    const float* W; int ldw;
    float* Y; int ldy;
    int M, N, K;
    const float* bias; int relu;
    const float* post_scale; const float* post_shift;
    unsigned* y_amax; int y_group_rows;
};

// A wave owns 8 rows x 8 columns and its 64 lanes split K (lane l takes k = 4l .. 4l+3 of every 256-wide step: every load is a
// coalesced 1 KiB row segment); the 64 lane-partial sums of each output are then combined by a halving butterfly (6 steps, 63
// shuffle+adds, fixed order) that leaves output q = 8 i + j in lane q.
__global__ __launch_bounds__(256) void gemm_skinny_kernel(SkinnyP p) {
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int r0 = blockIdx.y * 8, c0 = (blockIdx.x * 4 + wave) * 8;
    if (c0 >= p.N) return;                                   // wave-uniform
    const float* xr[8]; const float* wr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xr[i] = p.X + (size_t)min(r0 + i, p.M - 1) * p.ldx;
        wr[i] = p.W + (size_t)min(c0 + i, p.N - 1) * p.ldw;
    }
    float v[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) v[q] = 0.f;
    for (int k = 4 * l; k < p.K; k += 256) {
        float4 a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = *reinterpret_cast<const float4*>(xr[i] + k); b[i] = *reinterpret_cast<const float4*>(wr[i] + k); }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[8 * i + j] = fmaf(a[i].w, b[j].w, fmaf(a[i].z, b[j].z, fmaf(a[i].y, b[j].y, fmaf(a[i].x, b[j].x, v[8 * i + j]))));
    }
    // halving butterfly: at the step of lane bit m a lane keeps the half of its values selected by that bit and receives the
    // partner's copy of the same half
#define SKINNY_STEP(M_, N_)                                                                       \
    {                                                                                             \
        const bool hi = l & (M_);                                                                 \
        _Pragma("unroll") for (int q = 0; q < (N_) / 2; ++q)                                      \
            v[q] = (hi ? v[q + (N_) / 2] : v[q]) + __shfl_xor(hi ? v[q] : v[q + (N_) / 2], (M_), 64); \
    }
    SKINNY_STEP(32, 64) SKINNY_STEP(16, 32) SKINNY_STEP(8, 16) SKINNY_STEP(4, 8) SKINNY_STEP(2, 4) SKINNY_STEP(1, 2)
#undef SKINNY_STEP
    const int row = r0 + (l >> 3), col = c0 + (l & 7);
    float o = v[0];
    const bool ok = row < p.M && col < p.N;
    if (col < p.N) {
        if (p.bias) o += p.bias[col];
        if (p.relu) o = fmaxf(o, 0.f);
        if (p.post_scale) o = fmaf(o, p.post_scale[col], p.post_shift[col]);
    }
    if (ok) p.Y[(size_t)row * p.ldy + col] = o;
    if (p.y_amax) {                          // F16X2 range record of the output: one atomic per row and wave
        unsigned ab = ok ? (__float_as_uint(o) & 0x7fffffffu) : 0u;
#pragma unroll
        for (int s = 4; s >= 1; s >>= 1) ab = max(ab, (unsigned)__shfl_xor((int)ab, s, 64));
        if ((l & 7) == 0 && row < p.M && ab) atomicMax(&p.y_amax[row / p.y_group_rows], ab);
    }
}

}  // namespace

int ev2h_gemm_init() {       // per device (ev2h_init)
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES)););
    return EV2H_OK;
}

int ev2h_gemm_bf16(const ev2h_gemm_desc* d, ev2h_stream_t stream);

extern "C" int ev2h_gemm(const ev2h_gemm_desc* d, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(d && d->X && d->W && d->Y);
    EV2H_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0);
    EV2H_CHECK_ARG((d->K % 4) == 0 && (d->ldx % 4) == 0 && (d->ldw % 4) == 0);
    EV2H_CHECK_ARG(d->taps == 1 || d->taps == 3);
    GemmP p{};
    p.X = d->X; p.ldx = d->ldx; p.W = d->W; p.ldw = d->ldw; p.Y = d->Y; p.ldy = d->ldy;
    p.M = d->M; p.N = d->N; p.taps = d->taps; p.Kc = d->K; p.K = d->K * d->taps;
    p.rows_per_seq = d->rows_per_seq;
    if (d->taps == 3) EV2H_CHECK_ARG(d->rows_per_seq > 0 && d->M % d->rows_per_seq == 0 && d->ldw >= 3 * d->K);
    p.bias = d->bias; p.bias_group_rows = d->bias_group_rows; p.ldbias = d->ldbias;
    if (d->bias_group_rows > 0) EV2H_CHECK_ARG(d->bias && d->bias_group_rows % BM == 0);
    p.relu = d->relu; p.post_scale = d->post_scale; p.post_shift = d->post_shift;
    EV2H_CHECK_ARG((d->post_scale == nullptr) == (d->post_shift == nullptr));
    p.rowmax_rows = d->rowmax_rows;
    if (d->rowmax_rows) EV2H_CHECK_ARG(d->rowmax_rows == BM && d->M % BM == 0);
    if (d->K == 8 && d->taps == 1 && d->rowmax_rows == 0 && d->bias_group_rows == 0 && !d->post_scale && (d->N % 4) == 0 &&
        (d->ldy % 4) == 0 && (d->ldx % 4) == 0 && (d->ldw % 4) == 0) {      // (any M: a result must not depend on the batch size)
        // write-bound K = 8 layer (the layer-1 tables of the raw cloud): exact fp32 fma chains in every precision mode
        TableP t{};
        t.X = d->X; t.ldx = d->ldx; t.W = d->W; t.ldw = d->ldw; t.Y = d->Y; t.ldy = d->ldy; t.M = d->M; t.N = d->N;
        t.bias = d->bias; t.relu = d->relu;
        if ((d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_F16) && d->x_amax) {
            t.x_amax = d->x_amax; t.x_amax2 = d->x_amax2; t.x_group_rows = d->x_group_rows > 0 ? d->x_group_rows : 1;
            t.y_amax = d->y_amax; t.y_group_rows = d->y_group_rows > 0 ? d->y_group_rows : 1;
            t.y_scale = d->y_scale; t.y_bound_w = d->y_bound_w; t.y_bound_b = d->y_bound_b;
            if (t.y_scale) EV2H_CHECK_ARG(!t.y_amax || t.x_group_rows == t.y_group_rows);
        } else if ((d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_F16) && d->y_amax) {
            t.y_amax = d->y_amax; t.y_group_rows = d->y_group_rows > 0 ? d->y_group_rows : 1;
        }
        table_k8_kernel<<<ceil_div(d->M, TB_WAVES * TB_ROWS_PER_WAVE), 256, 0, (hipStream_t)stream>>>(t);
        EV2H_CHECK_LAUNCH();
        return EV2H_OK;
    }
    if (d->skinny) {
        EV2H_CHECK_ARG(d->taps == 1 && d->rowmax_rows == 0 && d->bias_group_rows == 0 && !d->y_scale);
        EV2H_CHECK_ARG(d->M <= 8 * 65535);                       // grid.y; the kernel is meant for M = number of windows anyway
        SkinnyP q{};
        q.X = d->X; q.ldx = d->ldx; q.W = d->W; q.ldw = d->ldw; q.Y = d->Y; q.ldy = d->ldy; q.M = d->M; q.N = d->N; q.K = d->K;
        q.bias = d->bias; q.relu = d->relu; q.post_scale = d->post_scale; q.post_shift = d->post_shift;
        if ((d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_F16) && d->y_amax) { q.y_amax = d->y_amax; q.y_group_rows = d->y_group_rows > 0 ? d->y_group_rows : 1; }
        gemm_skinny_kernel<<<dim3(ceil_div(d->N, 32), ceil_div(d->M, 8)), 256, 0, (hipStream_t)stream>>>(q);
        EV2H_CHECK_LAUNCH();
        return EV2H_OK;
    }
    if (d->precision != EV2H_PREC_F32) return ev2h_gemm_bf16(d, stream);
    const int tiles_m = ceil_div(d->M, BM);
    p.tiles_n = ceil_div(d->N, BN);
    p.nblk = tiles_m * p.tiles_n;
    gemm_nt_kernel<<<p.nblk, 256, GEMM_LDS_BYTES, (hipStream_t)stream>>>(p);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" int ev2h_transpose_logits(const float* logits_pm, int B, int N, float* logits_cm, size_t cm_window_stride, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(logits_pm && logits_cm && B > 0 && N > 0);
    EV2H_CHECK_ARG(cm_window_stride == 0 || cm_window_stride >= (size_t)4 * N);
    dim3 grid(ceil_div(N, 256), B);
    transpose_logits_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const float4*)logits_pm, N, logits_cm,
                                                                   cm_window_stride ? cm_window_stride : (size_t)4 * N);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}
