// Event window -> [5, N] tensor builder on the GPU (the step right before the hot path).
// Reference: /root/reference/src/Ev2Hands/dataset/evaluation_stream.py:187-225 (ERPCParser.__getitem__) and
// dataset/ev2hands_r.py:108-159: per-pixel accumulation of timestamp / positive / negative counts on the 346x260
// sensor with np.add.at, np.nonzero compaction in row-major order, t_avg = sum / count, resampling with replacement
// to N points, pc_normalize.
//
// np.add.at accumulates the float32 timestamp sum of a pixel in EVENT ORDER (each step: float64 add, round to float32),
// so a float atomicAdd scatter would not be bit-identical.  One workgroup per window instead sorts 32-bit keys (pixel << 15 | event index) with a bitonic network
// in LDS (<= 32768 events per window); equal-pixel events end up adjacent and in stream order, every run is summed
// sequentially in fp32 by the thread that owns its first element, and the runs come out already in np.nonzero order.
#include "common.hpp"
#include "ev2hands_hip.h"

namespace {

constexpr int EVW_THREADS = 1024;
constexpr int EVW_MAX_EVENTS = 32768;

__global__ __launch_bounds__(EVW_THREADS) void event_window_build_kernel(const double* __restrict__ events, const int32_t* __restrict__ offsets,
                                                                         int width, int height, int cap, int raw_time, int ev_stride,
                                                                         int32_t* __restrict__ uniq_count, float* __restrict__ uniq) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned* keys = reinterpret_cast<unsigned*>(smem_raw);
    __shared__ int s_part[EVW_THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int e0 = offsets[b], E = offsets[b + 1] - e0;
    if (E <= 0 || E > EVW_MAX_EVENTS) {
        if (tid == 0) uniq_count[b] = (E <= 0) ? 0 : -1;
        return;
    }
    int n = 1;
    while (n < E) n <<= 1;
    const double* ev = events + (size_t)e0 * ev_stride;
    for (int i = tid; i < n; i += EVW_THREADS) {
        unsigned k = 0xffffffffu;
        if (i < E) {
            const int x = (int)ev[(size_t)i * ev_stride + 0], y = (int)ev[(size_t)i * ev_stride + 1];     // .astype(np.int32): truncation
            const bool ok = x >= 0 && x < width && y >= 0 && y < height;
            k = ok ? ((unsigned)(y * width + x) << 15) | (unsigned)i : 0xfffffffeu;           // out-of-sensor events are dropped
        }
        keys[i] = k;
    }
    __syncthreads();
    // bitonic sort, ascending
    for (int k2 = 2; k2 <= n; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n; i += EVW_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned a = keys[i], c = keys[ixj];
                    const bool up = (i & k2) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    // run heads -> exclusive scan of head flags over contiguous per-thread chunks
    const int chunk = (n + EVW_THREADS - 1) / EVW_THREADS;
    const int lo = tid * chunk, hi = min(lo + chunk, n);
    int heads = 0;
    for (int i = lo; i < hi; ++i) {
        const unsigned k = keys[i];
        if (k < 0xfffffffeu && (i == 0 || (keys[i - 1] >> 15) != (k >> 15))) ++heads;
    }
    s_part[tid] = heads;
    __syncthreads();
    for (int off = 1; off < EVW_THREADS; off <<= 1) {           // Hillis-Steele inclusive scan
        const int v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int u = s_part[tid] - heads;                                 // exclusive prefix = index of this thread's first run
    const int total = s_part[EVW_THREADS - 1];
    const double t0 = raw_time ? 0.0 : ev[2];      // erpc.py accumulates the timestamps as they are, evaluation_stream.py minus the first
    float* out = uniq + (size_t)b * cap * 8;
    for (int i = lo; i < hi; ++i) {
        const unsigned k = keys[i];
        if (k >= 0xfffffffeu) break;
        const unsigned pix = k >> 15;
        if (i != 0 && (keys[i - 1] >> 15) == pix) continue;
        float tsum = 0.f;
        int cnt = 0, pos = 0;
        for (int q = i; q < n; ++q) {                            // the run may continue into the next thread's chunk
            const unsigned kq = keys[q];
            if ((kq >> 15) != pix) break;
            const int e = (int)(kq & 0x7fffu);
            // np.add.at(float32 grid, float64 t): each step adds in float64 and rounds the running sum to float32
            tsum = (float)((double)tsum + (ev[(size_t)e * ev_stride + 2] - t0));
            pos += (ev[(size_t)e * ev_stride + 3] == 1.0) ? 1 : 0;
            ++cnt;
        }
        if (u < cap) {
            float4* o = reinterpret_cast<float4*>(out + (size_t)u * 8);
            float tavg = __fdiv_rn(tsum, (float)cnt);
            if (raw_time) tavg = __fmul_rn(tavg, 1e-6f);         // erpc.py:191 "ns to ms", a float32 product
            o[0] = make_float4((float)(pix % width), (float)(pix / width), tavg, (float)pos);
            o[1] = make_float4((float)(cnt - pos), 0.f, 0.f, 0.f);
        }
        ++u;
    }
    if (tid == 0) uniq_count[b] = total;
}

// Ev2Hands-S (erpc.py:207-211): re-order a window's unique pixels by their mean time (np.argsort; pixels with exactly equal
// times -- whose order numpy leaves undefined -- stay in pixel order), subtract the first one's time, and pick the labels the
// reference picks: the per-EVENT label array indexed with the per-PIXEL sort positions (:209).
constexpr int EVS_MAX = 16384;
__global__ __launch_bounds__(EVW_THREADS) void event_window_timesort_kernel(const float* __restrict__ uniq_in, const int32_t* __restrict__ uniq_count, int cap,
                                                                            const double* __restrict__ events, int ev_stride, int label_col,
                                                                            const int32_t* __restrict__ offsets, float* __restrict__ uniq_out,
                                                                            int32_t* __restrict__ labels_out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int M = min(uniq_count[b], cap);
    if (M <= 0 || M > EVS_MAX) return;
    const float* tin = uniq_in + (size_t)b * cap * 8;
    int n = 1;
    while (n < M) n <<= 1;
    for (int i = tid; i < n; i += EVW_THREADS) {
        unsigned long long k = ~0ull;
        if (i < M) {
            unsigned u = __float_as_uint(tin[(size_t)i * 8 + 2]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // order-preserving map of IEEE floats to unsigned
            k = ((unsigned long long)u << 32) | (unsigned)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= n; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n; i += EVW_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], c = keys[ixj];
                    const bool up = (i & k2) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    const float tfirst = tin[(size_t)(unsigned)(keys[0] & 0xffffffffull) * 8 + 2];
    float* tout = uniq_out + (size_t)b * cap * 8;
    const double* ev = events ? events + (size_t)offsets[b] * ev_stride : nullptr;
    for (int j = tid; j < M; j += EVW_THREADS) {
        const unsigned src = (unsigned)(keys[j] & 0xffffffffull);
        float4 r0 = *reinterpret_cast<const float4*>(tin + (size_t)src * 8);
        const float4 r1 = *reinterpret_cast<const float4*>(tin + (size_t)src * 8 + 4);
        r0.z = __fsub_rn(r0.z, tfirst);
        *reinterpret_cast<float4*>(tout + (size_t)j * 8) = r0;
        *reinterpret_cast<float4*>(tout + (size_t)j * 8 + 4) = r1;
        if (labels_out) labels_out[(size_t)b * cap + j] = ev ? (int32_t)ev[(size_t)src * ev_stride + label_col] : 0;
    }
}

__global__ __launch_bounds__(256) void event_window_sample_kernel(const float* __restrict__ uniq, const int32_t* __restrict__ uniq_count, int cap,
                                                                  const int32_t* __restrict__ sample_idx, int N, int width, int height,
                                                                  float* __restrict__ out_cm, const int32_t* __restrict__ uniq_labels,
                                                                  int64_t* __restrict__ out_labels) {
    __shared__ float s_min[256], s_max[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int M = uniq_count[b];
    const float* tab = uniq + (size_t)b * cap * 8;
    const int32_t* idx = sample_idx + (size_t)b * N;
    float* o = out_cm + (size_t)b * 5 * N;
    float mn = INFINITY, mx = -INFINITY;
    for (int n = tid; n < N; n += 256) {
        int i = idx[n];
        i = (i < 0 || i >= M || i >= cap) ? 0 : i;
        const float t = tab[(size_t)i * 8 + 2];
        mn = fminf(mn, t);
        mx = fmaxf(mx, t);
    }
    s_min[tid] = mn; s_max[tid] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { s_min[tid] = fminf(s_min[tid], s_min[tid + off]); s_max[tid] = fmaxf(s_max[tid], s_max[tid + off]); }
        __syncthreads();
    }
    const float tmin = s_min[0], tmax = s_max[0];
    const float range = __fsub_rn(tmax, tmin);
    for (int n = tid; n < N; n += 256) {
        int i = idx[n];
        i = (i < 0 || i >= M || i >= cap) ? 0 : i;
        const float4 r0 = *reinterpret_cast<const float4*>(tab + (size_t)i * 8);
        const float neg = tab[(size_t)i * 8 + 4];
        // pc_normalize: x /= W; y /= H; xy = 2*xy - 1; t = 2*((t - tmin)/(tmax - tmin)) - 1   (float32 ops, no fma)
        o[0 * (size_t)N + n] = __fsub_rn(__fmul_rn(2.f, __fdiv_rn(r0.x, (float)width)), 1.f);
        o[1 * (size_t)N + n] = __fsub_rn(__fmul_rn(2.f, __fdiv_rn(r0.y, (float)height)), 1.f);
        o[2 * (size_t)N + n] = __fsub_rn(__fmul_rn(2.f, __fdiv_rn(__fsub_rn(r0.z, tmin), range)), 1.f);
        o[3 * (size_t)N + n] = r0.w;
        o[4 * (size_t)N + n] = neg;
        if (out_labels) out_labels[(size_t)b * N + n] = uniq_labels ? (int64_t)uniq_labels[(size_t)b * cap + i] : 0;
    }
}

}  // namespace

extern "C" int ev2h_event_window_build(const double* events, int ev_stride, const int32_t* offsets, int B, int width, int height, int cap,
                                       int raw_time, int32_t* uniq_count, float* uniq, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(events && offsets && uniq_count && uniq && ev_stride >= 4);
    EV2H_CHECK_ARG(B > 0 && width > 0 && height > 0 && width * height <= (1 << 17) && cap > 0);
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(event_window_build_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, EVW_MAX_EVENTS * 4)););
    event_window_build_kernel<<<B, EVW_THREADS, EVW_MAX_EVENTS * 4, (hipStream_t)stream>>>(events, offsets, width, height, cap, raw_time, ev_stride, uniq_count, uniq);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" int ev2h_event_window_timesort(const float* uniq_in, const int32_t* uniq_count, int cap, const double* events, int ev_stride,
                                          int label_col, const int32_t* offsets, int B, float* uniq_out, int32_t* labels_out,
                                          ev2h_stream_t stream) {
    EV2H_CHECK_ARG(uniq_in && uniq_count && uniq_out && uniq_in != uniq_out && B > 0 && cap > 0 && cap <= EVS_MAX);
    EV2H_CHECK_ARG(!labels_out || !events || (offsets && ev_stride > label_col && label_col >= 0));
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(event_window_timesort_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, EVS_MAX * 8)););
    event_window_timesort_kernel<<<B, EVW_THREADS, EVS_MAX * 8, (hipStream_t)stream>>>(uniq_in, uniq_count, cap, events, ev_stride, label_col,
                                                                                       offsets, uniq_out, labels_out);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" int ev2h_event_window_sample(const float* uniq, const int32_t* uniq_count, int cap, const int32_t* sample_idx, int B, int N,
                                        int width, int height, float* out_cm, const int32_t* uniq_labels, int64_t* out_labels,
                                        ev2h_stream_t stream) {
    EV2H_CHECK_ARG(uniq && uniq_count && sample_idx && out_cm);
    EV2H_CHECK_ARG(B > 0 && N > 0 && cap > 0 && width > 0 && height > 0);
    event_window_sample_kernel<<<B, 256, 0, (hipStream_t)stream>>>(uniq, uniq_count, cap, sample_idx, N, width, height, out_cm, uniq_labels,
                                                                   out_labels);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}
