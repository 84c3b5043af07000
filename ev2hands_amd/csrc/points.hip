// Point-set selection kernels of the Ev2Hands encoder for gfx950:
//   prep_points      [B,C,N] channel-major input -> point-major xyz(+|p|^2) and zero-padded features
//   fps              farthest point sampling            (reference: model/pointnet2_utils.py:63-84)
//   ball_query       first-K in-radius indices           (reference: model/pointnet2_utils.py:87-107)
//   three_nn_interp  3-NN inverse-distance interpolation (reference: model/pointnet2_utils.py:296-303)
// All discrete selections reproduce the reference's arithmetic forms exactly: squared norms as
// (x*x + y*y) + z*z with separate roundings, matmul-form distances as fma chains in k order
// (what MKL's K=3 sgemm produces), `d > r*r` in fp32, stable first-index tie breaks.
#include <cstdlib>
#include "common.hpp"
#include "ev2hands_hip.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sqnorm3(float x, float y, float z) {
    return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}

// ---------------------------------------------------------------------------------------- prep
__global__ __launch_bounds__(256) void prep_points_kernel(float* __restrict__ xyz_cm, int C, int N, int mhlnes,
                                                          float4* __restrict__ pts4, float4* __restrict__ feat8,
                                                          unsigned* __restrict__ amax) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    float* base = xyz_cm + (size_t)b * C * N;
    float f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = (c < C && n < N) ? base[(size_t)c * N + n] : 0.f;
    if (mhlnes && n < N) {
        // TEHNet.py:176-177: channel 2 is overwritten IN PLACE by the mean of channels 3..C-1
        float s = f[3];
        for (int c = 4; c < C; ++c) s = __fadd_rn(s, f[c]);
        s = s / (float)(C - 3);
        f[2] = s;
        base[(size_t)2 * N + n] = s;
    }
    if (amax) {                          // range record of the window's input channels (f16x2, ev2hands_hip.h "Range records")
        unsigned m = 0u;
#pragma unroll
        for (int c = 0; c < 8; ++c) m = max(m, __float_as_uint(f[c]) & 0x7fffffffu);
        m = wave_max_u32_dpp(m);
        if ((threadIdx.x & 63) == 0 && m) atomicMax(&amax[b], m);
    }
    if (n >= N) return;
    pts4[(size_t)b * N + n] = make_float4(f[0], f[1], f[2], sqnorm3(f[0], f[1], f[2]));
    feat8[((size_t)b * N + n) * 2 + 0] = make_float4(f[0], f[1], f[2], f[3]);
    feat8[((size_t)b * N + n) * 2 + 1] = make_float4(f[4], f[5], f[6], f[7]);
}

// ---------------------------------------------------------------------------------------- FPS
struct FpsJobs {
    int njobs;
    int S[3];
    const int64_t* init[3];
    int32_t* idx[3];
    float4* ctr[3];
    // chunked sampling [r6] (ev2h_fps_multi_chunk): this launch draws samples [s_begin, s_end) of every job (clipped to the job's S);
    // the running minima and the next start index travel between the launches of one sampling in `state`
    // (float [B][njobs][state_ld]: state_ld - 4 minima in the thread order p = j * THREADS + tid, then the index)
    int s_begin, s_end;
    float* state; int state_ld;
};

// THREADS = 256 with the window's points also in LDS (N <= 8192: the winner's coordinates are one LDS read away); windows beyond
// that -- the reference has no upper limit -- take THREADS = 1024, PPT <= 32 (N <= 32768) and read the winner from global memory
// (LDS_PTS = false: 32768 points would need 512 KB).
template <int PPT, int THREADS = 256, bool LDS_PTS = true>
__global__ __launch_bounds__(THREADS) void fps_kernel(const float4* __restrict__ pts4, int N, FpsJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* spts = reinterpret_cast<float4*>(smem_raw);
    constexpr int NW = THREADS / 64;
    __shared__ unsigned long long skey[2][NW];

    const int b = blockIdx.x;
    const int job = blockIdx.y;
    const int S = jobs.S[job];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4* src = pts4 + (size_t)b * N;

    // REG_PTS: a thread keeps its points' coordinates in registers.  Not at 32 points per thread in the 1024-thread form (N > 16 384):
    // 4 x 32 values do not fit the 128 registers of a 16-wave workgroup (the instantiation spilled 208 bytes into the serial loop) --
    // there only the running minima stay in registers and the coordinates are read again every step (coalesced, L2-resident).
#ifdef EV2H_FPS_REG_PTS_ALL      // (build switch for the A/B: the old form, spilling)
    constexpr bool REG_PTS = true;
#else
    constexpr bool REG_PTS = !(PPT >= 32 && THREADS >= 1024);
#endif
    float px[REG_PTS ? PPT : 1], py[REG_PTS ? PPT : 1], pz[REG_PTS ? PPT : 1], md[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = j * THREADS + tid;
        if constexpr (REG_PTS || LDS_PTS) {
            float4 v = (p < N) ? src[p] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (LDS_PTS && p < N) spts[p] = v;
            if constexpr (REG_PTS) { px[j] = v.x; py[j] = v.y; pz[j] = v.z; }
        }
        md[j] = 1e10f;
    }
    __syncthreads();
    int tid_v = tid;          // (!REG_PTS: laundered once per sampling step, so that the 32 load offsets are not hoisted out of the loop and spilled)
    auto coord = [&](int j, float& x, float& y, float& z) {
        if constexpr (REG_PTS) { x = px[j]; y = py[j]; z = pz[j]; }
        else {
            // (clamped index instead of a guarded load: no branch per point, one 32-bit offset per load; lanes past the window's end
            //  never enter the argmax -- the key comparison has its own p < N)
            const unsigned pc = (unsigned)min(j * THREADS + tid_v, N - 1);
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(src) + (size_t)pc * sizeof(float4));
            x = v.x; y = v.y; z = v.z;
        }
    };

    int far = (int)jobs.init[job][b];
    int32_t* oidx = jobs.idx[job] + (size_t)b * S;
    float4* octr = jobs.ctr[job] + (size_t)b * S;
    const int i_begin = min(jobs.s_begin, S), i_end = min(jobs.s_end, S);
    if (i_begin >= i_end) return;                        // (workgroup-uniform: this job was finished by an earlier chunk)
    float* st = jobs.state ? jobs.state + ((size_t)b * jobs.njobs + job) * jobs.state_ld : nullptr;
    if (i_begin > 0) {                                   // resume: the minima as the previous chunk left them
#pragma unroll
        for (int j = 0; j < PPT; ++j) md[j] = st[j * THREADS + tid];
        far = reinterpret_cast<const int*>(st)[jobs.state_ld - 4];
    }

    for (int i = i_begin; i < i_end; ++i) {
        if constexpr (!REG_PTS) asm volatile("" : "+v"(tid_v));
        const float4 c = LDS_PTS ? spts[far] : src[far];
        if (tid == 0) { oidx[i] = far; octr[i] = c; }
        unsigned long long best = 0ull;
        if constexpr (PPT >= 2) {
            // two points per packed fp32 instruction (v_pk_add_f32 / v_pk_mul_f32: same IEEE results, half the issue slots);
            // the kernel is VALU-throughput bound (three workgroups share a CU), so this is where the time goes
            const f32x2 cx = {c.x, c.x}, cy = {c.y, c.y}, cz = {c.z, c.z};
#pragma unroll
            for (int j = 0; j < PPT; j += 2) {
                float x0, y0, z0, x1, y1, z1;
                coord(j, x0, y0, z0);
                coord(j + 1, x1, y1, z1);
                const f32x2 dx = f32x2{x0, x1} - cx, dy = f32x2{y0, y1} - cy, dz = f32x2{z0, z1} - cz;
                const f32x2 d = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int p = (j + e) * THREADS + tid;
                    if (d[e] < md[j + e]) md[j + e] = d[e];
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(md[j + e]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)p);
                    if (p < N && key > best) best = key;
                }
                if constexpr (!REG_PTS) {
                    if ((j & 6) == 6) __builtin_amdgcn_sched_barrier(0);      // eight coordinate loads in flight at a time, not all 32
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int p = j * THREADS + tid;
                float x0, y0, z0;
                coord(j, x0, y0, z0);
                const float dx = __fsub_rn(x0, c.x), dy = __fsub_rn(y0, c.y), dz = __fsub_rn(z0, c.z);
                const float d = sqnorm3(dx, dy, dz);
                if (d < md[j]) md[j] = d;
                // key: larger distance first, then smaller index (torch.max returns the first maximum)
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(md[j]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)p);
                if (p < N && key > best) best = key;
            }
        }
        best = wave_max_u64_dpp(best);
        if (lane == 0) skey[i & 1][wave] = best;
        __syncthreads();
        unsigned long long k0 = skey[i & 1][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) { const unsigned long long kw = skey[i & 1][w]; k0 = k0 > kw ? k0 : kw; }
        far = (int)(0xffffffffu - (unsigned)(k0 & 0xffffffffull));
    }
    if (i_end < S) {                                     // to be continued by the next chunk
#pragma unroll
        for (int j = 0; j < PPT; ++j) st[j * THREADS + tid] = md[j];
        if (tid == 0) reinterpret_cast<int*>(st)[jobs.state_ld - 4] = far;
    }
}

// ---------------------------------------------------------------------------------------- ball query
struct BallArgs {
    int nrad;
    float r2[3];
    int K[3];
    int32_t* gidx[3];
    int32_t* cnt;   // [B][S][nrad] (may be null)
    int cpw;        // centroids per workgroup (multiple of 4: one per wave and round)
    int nchunk;     // workgroups per window = ceil(S / cpw); the grid is 1-D: nchunk * B workgroups
    int s_off, s_cnt;   // the centroids [s_off, s_off + s_cnt) of every window (ev2h_ball_query_range; the whole set: 0, S)
};

constexpr int BALL_CTR_PER_WG = 4, BALL_UNR = 4;      // one centroid per wave

// The window's points are read through L2 (coalesced 1 KiB per wave and step), never staged in LDS: a scan is a chain of dependent
// round trips with a data-dependent early exit, and what hides it is many short waves, not a copy of the points per workgroup.
// Round 4 moved the windows above 2048 points to this form (the 128 KB copy of an 8192-point window left one workgroup per CU: 3x);
// round 5 (XCD-aware grid, one centroid per wave) measured it ahead at 2048 points as well (+0.4 .. 1 % of the step, same-box
// builds, profiles/r5_ab_ball_l2_n2048.txt), so the LDS form is gone.
__global__ __launch_bounds__(256) void ball_query_kernel(const float4* __restrict__ pts4, const float4* __restrict__ ctr4,
                                                         int N, int S, BallArgs a) {
    // XCD-aware 1-D grid: the dispatcher deals workgroups to the 8 XCDs round-robin; xcd_remap gives every XCD a CONTIGUOUS range
    // of (window, chunk) pairs, so that the workgroups that scan one window's points sit on one XCD and find them in its L2.
    // (A [chunk, window] 2-D grid spread every window over all 8 L2s: at 128 windows of 8192 points each L2 saw all 16.8 MB of
    // points and the launch fetched 2.5 GB -- 150 x its input -- profiles/r4_pmc_hbm_traffic_n8192_f16x2.json.)
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / a.nchunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4* src = pts4 + (size_t)b * N;
    (void)tid;
    const int s_begin = a.s_off + (L - b * a.nchunk) * a.cpw;
    for (int s = s_begin + wave; s < s_begin + a.cpw && s < a.s_off + a.s_cnt; s += 4) {
        const float4 c = ctr4[(size_t)b * S + s];
        int cnt[3] = {0, 0, 0};
        int first[3] = {0, 0, 0};
        int32_t* gbase[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) gbase[i] = (i < a.nrad) ? a.gidx[i] + ((size_t)b * S + s) * a.K[i] : nullptr;
        // BALL_UNR chunks of 64 points per step: the distance evaluations and LDS reads of a step are independent, the
        // scalar bookkeeping (counts, exit test) is paid once per step; slots are still filled in ascending point order
        for (int base = 0; base < N; base += 64 * BALL_UNR) {
            float d[BALL_UNR];
#pragma unroll
            for (int u = 0; u < BALL_UNR; ++u) {
                const int p = base + 64 * u + lane;
                const float4 q = src[p < N ? p : N - 1];
                // square_distance (pointnet2_utils.py:37-39): -2*(c.q) + |c|^2 + |q|^2, dot as an fma chain
                const float dot = __fmaf_rn(c.z, q.z, __fmaf_rn(c.y, q.y, __fmul_rn(c.x, q.x)));
                d[u] = __fadd_rn(__fadd_rn(__fmul_rn(-2.f, dot), c.w), q.w);
            }
            bool done = true;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i < a.nrad && cnt[i] < a.K[i]) {       // wave-uniform: a radius whose K slots are full costs nothing more
#pragma unroll
                    for (int u = 0; u < BALL_UNR; ++u) {
                        const int p = base + 64 * u + lane;
                        const bool in = (p < N) && !(d[u] > a.r2[i]);
                        const unsigned long long m = __ballot(in);
                        if (cnt[i] == 0 && m != 0ull) first[i] = base + 64 * u + __ffsll((long long)m) - 1;
                        // slot = neighbours found so far + set bits below this lane (v_mbcnt)
                        const int pos = cnt[i] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (in && pos < a.K[i]) gbase[i][pos] = p;
                        cnt[i] += __popcll(m);
                    }
                    done = done && (cnt[i] >= a.K[i]);
                }
            }
            if (done) break;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < a.nrad) {
                const int have = cnt[i] < a.K[i] ? cnt[i] : a.K[i];
                for (int pos = have + lane; pos < a.K[i]; pos += 64)
                    a.gidx[i][((size_t)b * S + s) * a.K[i] + pos] = first[i];
                if (a.cnt && lane == 0) a.cnt[((size_t)b * S + s) * a.nrad + i] = have;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- 3-NN interpolation
constexpr int NN_PTS_PER_WG = 256;

__global__ __launch_bounds__(256) void three_nn_interp_kernel(const float4* __restrict__ pts1, const float4* __restrict__ pts2,
                                                              int N1, int N2, const float* __restrict__ feat2, int ldf2, int D,
                                                              float* __restrict__ out, int ldo,
                                                              int32_t* __restrict__ nn_idx, float* __restrict__ nn_w,
                                                              unsigned* __restrict__ amax, int ppw) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* s2 = reinterpret_cast<float4*>(smem_raw);                 // [N2]
    int* sidx = reinterpret_cast<int*>(s2 + N2);                       // [256][3]
    float* sw = reinterpret_cast<float*>(sidx + NN_PTS_PER_WG * 3);    // [256][3]
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int p = tid; p < N2; p += 256) s2[p] = pts2[(size_t)b * N2 + p];
    __syncthreads();

    // ppw = query points per workgroup (256, or 64 when the launch would otherwise leave most CUs idle: a few windows at a time)
    const int n = blockIdx.x * ppw + tid;
    if (tid < ppw && n < N1) {
        const float4 q = pts1[(size_t)b * N1 + n];
        float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
        int i0 = 0, i1 = 0, i2 = 0;
        for (int s = 0; s < N2; ++s) {
            const float4 c = s2[s];
            // src = xyz1 (q), dst = xyz2 (c): -2*dot + |q|^2 + |c|^2
            const float dot = __fmaf_rn(q.z, c.z, __fmaf_rn(q.y, c.y, __fmul_rn(q.x, c.x)));
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(-2.f, dot), q.w), c.w);
            if (d < d2) {               // strict: equal distances keep the earlier index (stable sort)
                if (d < d1) {
                    d2 = d1; i2 = i1;
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = s; }
                    else        { d1 = d;  i1 = s; }
                } else { d2 = d; i2 = s; }
            }
        }
        const float r0 = __fdiv_rn(1.0f, __fadd_rn(d0, 1e-8f));
        const float r1 = __fdiv_rn(1.0f, __fadd_rn(d1, 1e-8f));
        const float r2 = __fdiv_rn(1.0f, __fadd_rn(d2, 1e-8f));
        const float norm = __fadd_rn(__fadd_rn(r0, r1), r2);
        const float w0 = __fdiv_rn(r0, norm), w1 = __fdiv_rn(r1, norm), w2 = __fdiv_rn(r2, norm);
        sidx[tid * 3 + 0] = i0; sidx[tid * 3 + 1] = i1; sidx[tid * 3 + 2] = i2;
        sw[tid * 3 + 0] = w0; sw[tid * 3 + 1] = w1; sw[tid * 3 + 2] = w2;
        if (nn_idx) {
            int32_t* o = nn_idx + ((size_t)b * N1 + n) * 3;
            o[0] = i0; o[1] = i1; o[2] = i2;
        }
        if (nn_w) {
            float* o = nn_w + ((size_t)b * N1 + n) * 3;
            o[0] = w0; o[1] = w1; o[2] = w2;
        }
    }
    __syncthreads();
    if (!out) return;
    // each wave interpolates 64 of the block's points, lanes across channels (float4)
    const int D4 = D >> 2;
    float am = 0.f;                       // max |value| written by this lane (range record)
    // four points per iteration: their twelve row loads are in flight together (one point at a time left a wave waiting for a
    // dependent global load per point: 44 us for fp2's 2-workgroup launch at one window, tools/debug/latency_timeline.py)
    const int pw = ppw >> 2;             // points per wave
#pragma unroll 1
    for (int t0 = wave * pw; t0 < wave * pw + pw; t0 += 4) {
        if (blockIdx.x * ppw + t0 >= N1) break;
        for (int c = lane; c < D4; c += 64) {
            float4 a[4], bq[4], cq[4];
            float w0[4], w1[4], w2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u;
                const int j0 = sidx[t * 3 + 0], j1 = sidx[t * 3 + 1], j2 = sidx[t * 3 + 2];
                w0[u] = sw[t * 3 + 0]; w1[u] = sw[t * 3 + 1]; w2[u] = sw[t * 3 + 2];
                const bool live = blockIdx.x * ppw + t < N1;          // (sidx / sw of a dead slot were never written)
                a[u] = reinterpret_cast<const float4*>(feat2 + ((size_t)b * N2 + (live ? j0 : 0)) * ldf2)[c];
                bq[u] = reinterpret_cast<const float4*>(feat2 + ((size_t)b * N2 + (live ? j1 : 0)) * ldf2)[c];
                cq[u] = reinterpret_cast<const float4*>(feat2 + ((size_t)b * N2 + (live ? j2 : 0)) * ldf2)[c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int nn = blockIdx.x * ppw + t0 + u;
                if (nn >= N1) break;
                float4 r;
                // torch.sum(dim=2) of the three weighted rows: (a*w0 + b*w1) + c*w2, products rounded first
                r.x = __fadd_rn(__fadd_rn(__fmul_rn(a[u].x, w0[u]), __fmul_rn(bq[u].x, w1[u])), __fmul_rn(cq[u].x, w2[u]));
                r.y = __fadd_rn(__fadd_rn(__fmul_rn(a[u].y, w0[u]), __fmul_rn(bq[u].y, w1[u])), __fmul_rn(cq[u].y, w2[u]));
                r.z = __fadd_rn(__fadd_rn(__fmul_rn(a[u].z, w0[u]), __fmul_rn(bq[u].z, w1[u])), __fmul_rn(cq[u].z, w2[u]));
                r.w = __fadd_rn(__fadd_rn(__fmul_rn(a[u].w, w0[u]), __fmul_rn(bq[u].w, w1[u])), __fmul_rn(cq[u].w, w2[u]));
                reinterpret_cast<float4*>(out + ((size_t)b * N1 + nn) * ldo)[c] = r;
                am = fmaxf(fmaxf(am, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
            }
        }
    }
    if (amax) {                          // range record of the interpolated rows (f16x2)
        const unsigned m = wave_max_u32_dpp(__float_as_uint(am));
        if (lane == 0 && m) atomicMax(&amax[b], m);
    }
}

}  // namespace

// ======================================================================================== C ABI
extern "C" int ev2h_prep_points(float* xyz_cm, int B, int C, int N, int mhlnes, float* pts4, float* feat8, uint32_t* amax,
                                ev2h_stream_t stream) {
    EV2H_CHECK_ARG(xyz_cm && pts4 && feat8);
    EV2H_CHECK_ARG(B > 0 && N > 0 && C >= 4 && C <= 8);
    dim3 grid(ceil_div(N, 256), B);
    prep_points_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(xyz_cm, C, N, mhlnes, (float4*)pts4, (float4*)feat8, amax);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

// floats of sampling state per (window, job) for ev2h_fps_multi_chunk: the padded thread layout of the kernel that samples N points + the index
size_t ev2h_fps_state_ld(int N) {
    const int threads = N <= 2048 ? 256 : (N <= 8192 ? 512 : 1024);
    const int ppt = N <= 256 ? 1 : N <= 512 ? 2 : N <= 1024 ? 4 : N <= 2048 ? 8 : N <= 4096 ? 8 : N <= 8192 ? 16 : N <= 16384 ? 16 : 32;
    return (size_t)threads * ppt + 4;
}

int ev2h_fps_multi_chunk(const float* pts4, int B, int N, int njobs, const int* S, const int64_t* const* init, int32_t* const* idx, float* const* ctr4,
                         int s_begin, int s_end, float* state, ev2h_stream_t stream);

extern "C" int ev2h_fps_multi(const float* pts4, int B, int N, int njobs, const int* S, const int64_t* const* init,
                              int32_t* const* idx, float* const* ctr4, ev2h_stream_t stream) {
    return ev2h_fps_multi_chunk(pts4, B, N, njobs, S, init, idx, ctr4, 0, 0x7fffffff, nullptr, stream);
}

// internal (forward.hip) [r6]: samples [s_begin, s_end) of every job in this launch; `state` (B * njobs * ev2h_fps_state_ld(N) floats)
// carries a sampling from one launch to the next -- the same sequence of arg-max steps as one launch, identical indices.  A small
// batch leaves most of the chip idle for the 512 dependent steps of enc.sa1's sampling: in chunks, the ball query and the fused set
// abstraction of the centroids already drawn run beside the rest of the sampling (ev2h_forward).
int ev2h_fps_multi_chunk(const float* pts4, int B, int N, int njobs, const int* S, const int64_t* const* init, int32_t* const* idx, float* const* ctr4,
                         int s_begin, int s_end, float* state, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(pts4 && S && init && idx && ctr4);
    EV2H_CHECK_ARG(B > 0 && N > 0 && N <= 32768 && njobs >= 1 && njobs <= 3);
    EV2H_CHECK_ARG(s_begin >= 0 && s_end > s_begin);
    FpsJobs jobs{};
    jobs.njobs = njobs;
    jobs.s_begin = s_begin; jobs.s_end = s_end; jobs.state = state; jobs.state_ld = (int)ev2h_fps_state_ld(N);
    for (int j = 0; j < njobs; ++j) {
        EV2H_CHECK_ARG(S[j] > 0 && init[j] && idx[j] && ctr4[j]);
        jobs.S[j] = S[j];
        jobs.init[j] = init[j];
        jobs.idx[j] = idx[j];
        jobs.ctr[j] = (float4*)ctr4[j];
        EV2H_CHECK_ARG(state || (s_begin == 0 && s_end >= S[j]));      // a sampling that spans launches needs its state buffer
    }
    dim3 grid(B, njobs);
    const size_t lds = (size_t)N * sizeof(float4);
    hipStream_t st = (hipStream_t)stream;
    const float4* p = (const float4*)pts4;
    if (N > 2048 && N <= 8192) {   // > 32 KiB of points: raise the dynamic-LDS limit once
        static PerDevice attr_set{};
        EV2H_ONCE_PER_DEVICE(attr_set,
            EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<8, 512, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 16));
            EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<16, 512, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16)););
    }
    if (N <= 256) fps_kernel<1><<<grid, 256, lds, st>>>(p, N, jobs);
    else if (N <= 512) fps_kernel<2><<<grid, 256, lds, st>>>(p, N, jobs);
    else if (N <= 1024) fps_kernel<4><<<grid, 256, lds, st>>>(p, N, jobs);
    else if (N <= 2048) fps_kernel<8><<<grid, 256, lds, st>>>(p, N, jobs);
    // N > 2048: the staged points (64 / 128 KB) leave two / one workgroup per CU, and with 16 / 32 points per thread the distance update
    // is most of a sampling step -- eight waves per window halve it (N = 2048 keeps four: there the exchange and the barrier dominate
    // and sixteen waves measured slower; sixteen waves at 8192 points [r6]: 16 x 8192 7 135 against 7 515 windows/s,
    // profiles/r6_ab_fps1024_refuted.txt).  Same maxima, same tie-break: identical indices.
    else if (N <= 4096) fps_kernel<8, 512, true><<<grid, 512, lds, st>>>(p, N, jobs);
    else if (N <= 8192) fps_kernel<16, 512, true><<<grid, 512, lds, st>>>(p, N, jobs);
    else if (N <= 16384) fps_kernel<16, 1024, false><<<grid, 1024, 0, st>>>(p, N, jobs);      // beyond the LDS-resident sizes
    else fps_kernel<32, 1024, false><<<grid, 1024, 0, st>>>(p, N, jobs);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" int ev2h_fps(const float* pts4, int B, int N, int S, const int64_t* init, int32_t* idx, float* ctr4,
                        ev2h_stream_t stream) {
    return ev2h_fps_multi(pts4, B, N, 1, &S, &init, &idx, &ctr4, stream);
}

int ev2h_ball_query_range(const float* pts4, const float* ctr4, int B, int N, int S, int s_off, int s_cnt, int nrad, const double* radius,
                          const int* nsample, int32_t* const* gidx, int32_t* cnt, ev2h_stream_t stream);

extern "C" int ev2h_ball_query(const float* pts4, const float* ctr4, int B, int N, int S, int nrad, const double* radius,
                               const int* nsample, int32_t* const* gidx, int32_t* cnt, ev2h_stream_t stream) {
    return ev2h_ball_query_range(pts4, ctr4, B, N, S, 0, S, nrad, radius, nsample, gidx, cnt, stream);
}

// internal (forward.hip) [r6]: the same for the centroids [s_off, s_off + s_cnt) of every window (arrays indexed as for the whole set)
int ev2h_ball_query_range(const float* pts4, const float* ctr4, int B, int N, int S, int s_off, int s_cnt, int nrad, const double* radius,
                          const int* nsample, int32_t* const* gidx, int32_t* cnt, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(pts4 && ctr4 && radius && nsample && gidx);
    EV2H_CHECK_ARG(B > 0 && N > 0 && N <= 32768 && S > 0 && nrad >= 1 && nrad <= 3 && s_off >= 0 && s_cnt > 0 && s_off + s_cnt <= S);
    BallArgs a{};
    a.nrad = nrad;
    for (int i = 0; i < nrad; ++i) {
        EV2H_CHECK_ARG(gidx[i] && nsample[i] > 0);
        // `radius ** 2` is evaluated in Python double precision, then rounded to fp32 for the comparison (pointnet2_utils.py:102)
        a.r2[i] = (float)(radius[i] * radius[i]);
        a.K[i] = nsample[i];
        a.gidx[i] = gidx[i];
    }
    a.cnt = cnt;
    a.cpw = BALL_CTR_PER_WG;
    a.s_off = s_off; a.s_cnt = s_cnt;
    a.nchunk = ceil_div(s_cnt, a.cpw);
    ball_query_kernel<<<dim3((unsigned)a.nchunk * (unsigned)B), 256, 0, (hipStream_t)stream>>>((const float4*)pts4, (const float4*)ctr4, N, S, a);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" int ev2h_three_nn_interp(const float* pts1_4, const float* pts2_4, int B, int N1, int N2, const float* feat2,
                                    int ldf2, int D, float* out, int ldo, int32_t* nn_idx, float* nn_w, uint32_t* out_amax,
                                    ev2h_stream_t stream) {
    EV2H_CHECK_ARG(pts1_4 && pts2_4);
    EV2H_CHECK_ARG(B > 0 && N1 > 0 && N2 >= 3 && N2 <= 4096);
    if (out) EV2H_CHECK_ARG(feat2 && D > 0 && (D % 4) == 0 && (ldf2 % 4) == 0 && (ldo % 4) == 0);
    const int ppw = ((long)ceil_div(N1, NN_PTS_PER_WG) * B < 128) ? 64 : NN_PTS_PER_WG;
    dim3 grid(ceil_div(N1, ppw), B);
    const size_t lds = (size_t)N2 * sizeof(float4) + NN_PTS_PER_WG * 3 * (sizeof(int) + sizeof(float));
    three_nn_interp_kernel<<<grid, 256, lds, (hipStream_t)stream>>>((const float4*)pts1_4, (const float4*)pts2_4, N1, N2, feat2,
                                                                    ldf2, D, out, ldo, nn_idx, nn_w, out_amax, ppw);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}
