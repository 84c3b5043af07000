// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the Ev2Hands hot path.
// Wavefront = 64 lanes everywhere; no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define EV2H_OK 0
#define EV2H_ERR_ARG 1
#define EV2H_ERR_HIP 2
#define EV2H_ERR_WORKSPACE 3

void ev2h_set_error(const char* fmt, ...);

#define EV2H_CHECK_ARG(cond)                                                        \
    do {                                                                            \
        if (!(cond)) {                                                              \
            ev2h_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond);   \
            return EV2H_ERR_ARG;                                                    \
        }                                                                           \
    } while (0)

#define EV2H_CHECK_HIP(expr)                                                                     \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            ev2h_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return EV2H_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

#define EV2H_CHECK_LAUNCH() EV2H_CHECK_HIP(hipGetLastError())

// Per-device, thread-safe one-time setup (kernel attributes such as the dynamic-LDS limit are per device, and a host may drive
// several devices from several threads).  `slot` is a static PerDevice object at the call site; the setup itself is idempotent,
// so two threads racing through it is harmless.
#include <atomic>
constexpr int EV2H_MAX_DEV = 16;
struct PerDevice {
    std::atomic<int> v[EV2H_MAX_DEV];
    std::atomic<int>& cur() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return v[(dev >= 0 && dev < EV2H_MAX_DEV) ? dev : 0];
    }
};
#define EV2H_ONCE_PER_DEVICE(slot, ...)                 \
    do {                                                \
        std::atomic<int>& f__ = (slot).cur();           \
        if (!f__.load(std::memory_order_acquire)) {     \
            __VA_ARGS__;                                \
            f__.store(1, std::memory_order_release);    \
        }                                               \
    } while (0)

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact f32 (fmaf chain in k order).
//   A: lane l holds A[i = l&31][k = l>>5];  B: lane l holds B[k = l>>5][j = l&31]
//   D: lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of D register r for lane-half h
__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// XCD-aware bijective remap of a 1-D block id: blocks that the dispatcher places on the same XCD
// (id % 8) get a contiguous range of logical ids, so neighbouring work shares that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// wave-wide unsigned max with DPP row shifts / row broadcasts (6 short-latency steps instead of 6 ds_bpermute round trips);
// result broadcast from lane 63
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned v) {
#define EV2H_DPP_MAX(ctrl, rmask) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false))
    EV2H_DPP_MAX(0x111, 0xf);      // row_shr:1
    EV2H_DPP_MAX(0x112, 0xf);      // row_shr:2
    EV2H_DPP_MAX(0x114, 0xf);      // row_shr:4
    EV2H_DPP_MAX(0x118, 0xf);      // row_shr:8   -> lane 15 of every 16-lane row holds the row's max
    EV2H_DPP_MAX(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    EV2H_DPP_MAX(0x143, 0xc);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's max
#undef EV2H_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// max of (hi, lo) pairs ordered by hi, then lo: two 32-bit DPP reductions
__device__ __forceinline__ unsigned long long wave_max_u64_dpp(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mhi = wave_max_u32_dpp(hi);
    const unsigned mlo = wave_max_u32_dpp(hi == mhi ? lo : 0u);
    return ((unsigned long long)mhi << 32) | mlo;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffull), o, 64);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), o, 64);
        unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = (w > v) ? w : v;
    }
    return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
