// ev2h_forward: TEHNet.forward (/root/reference/src/Ev2Hands/model/TEHNet.py:168-197) as one
// in-order sequence of gfx950 kernels on a caller-provided stream and workspace.  No allocation,
// no host synchronisation, no device->host copies inside (hipGraph-capturable).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "planes.hpp"
#include "ev2hands_hip.h"

int ev2h_gemm_init();

// ---------------------------------------------------------------------------------------- errors / init
static thread_local char g_err[512] = "";

void ev2h_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ev2h_last_error(void) { return g_err; }
extern "C" int ev2h_abi_version(void) { return EV2H_ABI_VERSION; }

extern "C" void ev2h_struct_sizes(size_t out[8]) {
    out[0] = sizeof(ev2h_gemm_desc);
    out[1] = sizeof(ev2h_sa_desc);
    out[2] = sizeof(ev2h_sa_module);
    out[3] = sizeof(ev2h_weights);
    out[4] = sizeof(ev2h_mano_consts);
    out[5] = sizeof(ev2h_outputs);
    out[6] = sizeof(ev2h_fp_desc);
    out[7] = sizeof(ev2h_tensor_desc);
}

static struct SideCtx* side_ctx(void* caller_stream = nullptr);
static void side_open(struct SideCtx& c);
// per-device, thread-safe, idempotent (common.hpp: PerDevice).  Also creates the calling thread's side stream on the current device
// NOW: HIP multiplexes streams onto a few hardware queues, and two streams that share one run in order -- a host that is going to
// create many more streams (torch's stream pool, RCCL's) should call this first, so that the forward's side stream gets a hardware
// queue of its own instead of landing on the caller's (measured: the two-stream overlaps, ~5 % of the step, silently vanish).
extern "C" int ev2h_init(void) {
    (void)side_ctx();
    return ev2h_gemm_init();
}

// ---------------------------------------------------------------------------------------- profiling hook
// bench.py brackets ONE named launch site of ev2h_forward with caller-owned HIP events (recorded on the
// forward's own stream), cycling through n event pairs so that K timed steps give K samples.
// (per host thread: the thread that arms the hook is the one whose forwards are bracketed)
static thread_local struct {
    char tag[32];
    hipEvent_t* start;
    hipEvent_t* stop;
    int n;
    long calls;
} g_prof = {"", nullptr, nullptr, 0, 0};

extern "C" int ev2h_profile_set(const char* tag, void** start_events, void** stop_events, int n) {
    if (!tag || n <= 0 || !start_events || !stop_events) {
        g_prof.tag[0] = 0; g_prof.start = g_prof.stop = nullptr; g_prof.n = 0; g_prof.calls = 0;
        return EV2H_OK;
    }
    snprintf(g_prof.tag, sizeof(g_prof.tag), "%s", tag);
    g_prof.start = reinterpret_cast<hipEvent_t*>(start_events);
    g_prof.stop = reinterpret_cast<hipEvent_t*>(stop_events);
    g_prof.n = n;
    g_prof.calls = 0;
    return EV2H_OK;
}

static inline bool prof_hit(const char* tag) { return g_prof.n > 0 && !strcmp(tag, g_prof.tag); }
static inline void prof_begin(const char* tag, ev2h_stream_t st) {
    if (prof_hit(tag)) (void)hipEventRecord(g_prof.start[g_prof.calls % g_prof.n], (hipStream_t)st);
}
static inline void prof_end(const char* tag, ev2h_stream_t st) {
    if (prof_hit(tag)) { (void)hipEventRecord(g_prof.stop[g_prof.calls % g_prof.n], (hipStream_t)st); ++g_prof.calls; }
}

// ---------------------------------------------------------------------------------------- side stream
// The two MANO regressors are independent after the attention block, and their ball queries depend only on the
// sampled centroids.  They are forked onto one library-owned side stream per host thread (fork/join with events,
// hipGraph-capturable), so the small kernels of one hand (ball query, table GEMM, head GEMMs, MANO) overlap the
// MFMA-heavy kernels of the other and fill their tails: +1.5-2 % windows/s at B=256, outputs bit-identical
// (tests/test_gpu_forward.py::test_two_stream_fork_is_bit_identical).  EV2H_TWO_STREAMS=0 keeps everything on the caller's
// stream.  Kernels of the two hands then overlap in time, so bench.py brackets a launch site before the fork (sa2.1).
// One side stream (+ its events) per host thread AND per device: a second wrapper on another GPU in the same thread gets
// its own stream on that device.
struct SideCtx {
    hipStream_t stream = nullptr;
    static constexpr int NEV = 14;     // (10 .. 13: the chunks of enc.sa1's sampling)
    hipEvent_t ev[NEV] = {};
    int state = 0;               // 0 = not tried, 1 = ready, -1 = disabled
    void* owner = nullptr;       // the caller's stream this side stream serves (slot 0: the first caller's, claimed at its first forward)
    bool claimed = false;
    bool bound = false;          // ev2h_bind_stream has measured this pair (and replaced the stream if it shared the caller's hardware queue)
    unsigned long long last_use = 0;      // g_side_tick of the last forward / probe that looked this slot up (recycling, see side_ctx)
};
constexpr int EV2H_MAX_DEVICES = 16;
// [r6] One side stream PER CALLER STREAM (up to EV2H_SIDE_SLOTS per host thread and device): forwards that are in flight at the same
// time on different streams (ev2hands_amd/inflight.py, dist.GatherPipeline(inflight=K)) used to share ONE side stream -- harmless
// while it carried only the tails of a forward, but since enc.sa1's sampling runs there (chunked, ev2h_fps_multi_chunk) forward
// i + 1's sampling queued behind forward i's right-hand regressor and two forwards in flight bought nothing (16 x 8192: 7 557
// against 7 568 windows/s with one).  Slot 0 is the stream ev2h_init creates first (it wants a hardware queue of its own).
constexpr int EV2H_SIDE_SLOTS = 4;
static thread_local SideCtx g_side[EV2H_MAX_DEVICES][EV2H_SIDE_SLOTS];

static thread_local unsigned long long g_side_tick = 0;
static thread_local bool g_side_claim = false;      // set by ev2h_forward / the probe around side_ctx(): this call binds a slot to its caller stream
static inline bool caller_stream_claims(void*) { return g_side_claim; }
static thread_local int g_side_disabled = 0;      // ev2h_set_side_stream(0): run everything on the caller's stream (per host thread)

extern "C" int ev2h_set_side_stream(int enabled) {
    const int prev = !g_side_disabled;
    g_side_disabled = !enabled;
    return prev;
}

static SideCtx* side_ctx(void* caller_stream) {     // the side stream that serves `caller_stream` on the current device, or nullptr (single-stream mode)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= EV2H_MAX_DEVICES) return nullptr;
    int slot = 0;
    if (caller_stream || g_side[dev][0].claimed) {
        // the slot that already serves this caller stream, else the first free one; all taken: the slot that has not been looked up
        // for the longest time is RECYCLED if it has been idle for a while (a host that keeps making new streams -- one
        // InflightForward per request, torch's pool handing out other handles -- would otherwise be stuck with its first four
        // forever), else slot 0 is shared (correct, only serialised: more than four streams in rotation must not evict each other --
        // every eviction costs a probe).  Handing a side stream to a new owner is safe whatever it is still running: each forward
        // forks it by an event wait and joins it by an event before it returns, and the enqueue calls of one host thread do not interleave.
        int found = -1, free_ = -1, lru = 0;
        for (int i = 0; i < EV2H_SIDE_SLOTS; ++i) {
            if (g_side[dev][i].claimed && g_side[dev][i].owner == caller_stream) { found = i; break; }
            if (!g_side[dev][i].claimed && free_ < 0) free_ = i;
            if (g_side[dev][i].last_use < g_side[dev][lru].last_use) lru = i;
        }
        slot = found >= 0 ? found : (free_ >= 0 ? free_ : 0);
        if (found < 0 && free_ < 0 && caller_stream_claims(caller_stream) && g_side_tick - g_side[dev][lru].last_use >= 16) {
            slot = lru;
            g_side[dev][slot].claimed = false;          // re-claimed just below, for the new owner; measured again by ev2h_bind_stream
            g_side[dev][slot].bound = false;
        }
    }
    SideCtx& c = g_side[dev][slot];
    if (caller_stream_claims(caller_stream) && !c.claimed) { c.claimed = true; c.owner = caller_stream; }
    if (caller_stream_claims(caller_stream) && c.claimed && c.owner == caller_stream) c.last_use = ++g_side_tick;
    side_open(c);
    return c.state == 1 ? &c : nullptr;
}

static void side_open(SideCtx& c) {
    if (c.state == 0) {
        const char* e = getenv("EV2H_TWO_STREAMS");
        c.state = -1;
        // A NORMAL-priority, non-blocking stream, created as early as possible (ev2h_init).  Measured alternatives, 1-rank RCCL
        // process, B = 256 (profiles/r3_dist_overhead.txt): a low- or high-priority side stream (its own queue class): -12 %;
        // GPU_MAX_HW_QUEUES=8 with the side stream created first: -10 % (more hardware queues than the scheduler maps at once);
        // side stream created after torch's / RCCL's streams with the default 4 queues: -5 % (it shares the caller's queue).
        if (!(e && atoi(e) == 0) && hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) == hipSuccess) {
            bool ok = true;
            for (int i = 0; i < SideCtx::NEV; ++i) ok = ok && hipEventCreateWithFlags(&c.ev[i], hipEventDisableTiming) == hipSuccess;
            if (ok) c.state = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------- side-stream probe
namespace {
// spins for ~ticks of the constant-rate real-time counter (100 MHz on gfx950) without touching memory
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long r0 = wall_clock64();
    while (wall_clock64() - r0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

namespace {
// (time of one spin kernel on each of a and b at once) / (time of one on a alone): ~1.0-1.3 = concurrent, ~2 = the streams share a hardware queue
hipError_t probe_pair(hipStream_t a, hipStream_t b, int spin_us, float* ratio) {
    *ratio = 0.f;
    hipEvent_t e[6] = {};
    hipError_t err = hipSuccess;
    for (auto& x : e) if (err == hipSuccess) err = hipEventCreate(&x);
    int rate_khz = 100000, dev = 0;                         // wall_clock64 ticks per millisecond
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
    const unsigned long long ticks = (unsigned long long)rate_khz * (unsigned long long)spin_us / 1000ull;
    float one = 0.f, two = 0.f;
    for (int rep = 0; rep < 2 && err == hipSuccess; ++rep) {   // (first repetition: code load, queue wake-up)
        err = hipEventRecord(e[0], a);
        spin_kernel<<<1, 64, 0, a>>>(ticks);
        if (err == hipSuccess) err = hipEventRecord(e[1], a);
        if (err == hipSuccess) err = hipEventRecord(e[4], a);                   // fork exactly as ev2h_forward does
        if (err == hipSuccess) err = hipStreamWaitEvent(b, e[4], 0);
        if (err == hipSuccess) err = hipEventRecord(e[2], a);
        spin_kernel<<<1, 64, 0, a>>>(ticks);
        spin_kernel<<<1, 64, 0, b>>>(ticks);
        if (err == hipSuccess) err = hipEventRecord(e[5], b);
        if (err == hipSuccess) err = hipStreamWaitEvent(a, e[5], 0);
        if (err == hipSuccess) err = hipEventRecord(e[3], a);
        if (err == hipSuccess) err = hipStreamSynchronize(a);
    }
    if (err == hipSuccess) err = hipEventElapsedTime(&one, e[0], e[1]);
    if (err == hipSuccess) err = hipEventElapsedTime(&two, e[2], e[3]);
    for (auto& x : e) if (x) (void)hipEventDestroy(x);
    if (err == hipSuccess) *ratio = one > 0.f ? two / one : 0.f;
    return err;
}
constexpr float SERIALISED = 1.6f;          // concurrent pairs measure 1.0-1.3 (the second launch's latency), serialised ones 1.9-2.1
}  // namespace

extern "C" int ev2h_streams_concurrent(ev2h_stream_t a, ev2h_stream_t b, int spin_us, float* ratio) {
    EV2H_CHECK_ARG(ratio && spin_us > 0 && spin_us <= 100000);
    const hipError_t err = probe_pair((hipStream_t)a, (hipStream_t)b, spin_us, ratio);
    if (err != hipSuccess) { ev2h_set_error("ev2h_streams_concurrent: %s", hipGetErrorString(err)); return EV2H_ERR_HIP; }
    return EV2H_OK;
}

// [r6] HIP multiplexes a process's streams onto a few hardware queues (4 by default) and two streams that share one run IN ORDER, without
// any error.  Which queue a stream gets depends on what the process created before it (torch's pool of 32, RCCL's streams, other
// libraries): with the library's side stream created first and one forward at a time the default mapping works (ev2h_init), but
// with forwards in flight on several caller streams -- each with a side stream of its own -- no creation order is right for every
// host (measured in the 1-rank RCCL process, 16 x 8192, profiles/r6_side_slots_ab.txt: every slot created at ev2h_init: two in flight
// 9 010 windows/s but ONE in flight 5 980 instead of 7 400 and B = 256 -4 %; slots created at first use: one in flight 7 400, two
// 7 350 instead of 9 000).  So the mapping is MEASURED: ev2h_bind_stream probes candidate side streams against the caller's stream and
// against the streams this thread has bound before, and keeps the one that really runs beside them.
extern "C" int ev2h_bind_stream(ev2h_stream_t stream, int* info) {
    if (info) info[0] = info[1] = info[2] = 0;
    if (g_side_disabled) return EV2H_OK;
    int dev = 0;
    EV2H_CHECK_HIP(hipGetDevice(&dev));
    EV2H_CHECK_ARG(dev >= 0 && dev < EV2H_MAX_DEVICES);
    g_side_claim = true;
    SideCtx* side = side_ctx(stream);
    g_side_claim = false;
    if (!side) return EV2H_OK;                                  // single-stream mode: nothing to bind
    if (!(side->claimed && side->owner == stream)) return EV2H_OK;   // every slot taken: this stream shares slot 0 (serialised with its owner's tails, correct)
    if (side->bound) return EV2H_OK;                            // measured before: the cheap path of a call per forward
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return EV2H_OK;      // the probe synchronises: not now (stays unbound)
    // the streams bound before on this thread and device: a good side stream also stays out of THEIR way
    std::vector<hipStream_t> others;
    for (int i = 0; i < EV2H_SIDE_SLOTS; ++i) {
        SideCtx& o = g_side[dev][i];
        if (&o == side || !o.claimed || o.state != 1) continue;
        others.push_back((hipStream_t)o.owner);
        others.push_back(o.stream);
    }
    constexpr int SPIN_US = 40;
    hipError_t err = hipSuccess;
    auto score = [&](hipStream_t cand, float* own_ratio) {      // 100 if serialised with its own caller stream, + 1 per other stream it is serialised with
        int sc = 0;
        float r = 0.f;
        if (err == hipSuccess) err = probe_pair(st, cand, SPIN_US, &r);
        *own_ratio = r;
        if (r > SERIALISED) sc += 100;
        for (hipStream_t o : others) {
            float ro = 0.f;
            if (err == hipSuccess) err = probe_pair(o, cand, SPIN_US, &ro);
            if (ro > SERIALISED) ++sc;
        }
        return sc;
    };
    float best_ratio = 0.f;
    int best = score(side->stream, &best_ratio), tried = 1;
    std::vector<hipStream_t> rejected;
    while (err == hipSuccess && best > 0 && tried < 8) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
        ++tried;
        float r = 0.f;
        const int sc = score(cand, &r);
        if (err == hipSuccess && sc < best) { rejected.push_back(side->stream); side->stream = cand; best = sc; best_ratio = r; }
        else rejected.push_back(cand);
    }
    // (destroyed only now: a destroyed stream's queue slot would be handed to the next candidate)
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    if (err != hipSuccess) { ev2h_set_error("ev2h_bind_stream: %s", hipGetErrorString(err)); return EV2H_ERR_HIP; }
    side->bound = true;
    // (best > 0: more streams in flight than hardware queues -- two forwards and their side streams fill the default four.  Running
    //  such a caller stream WITHOUT a side stream was measured and is worse: 16 x 8192, three in flight, 8 470 against 9 030 windows/s.)
    if (info) { info[0] = tried; info[1] = (int)(best_ratio * 1000.f + 0.5f); info[2] = best % 100; }
    return EV2H_OK;
}

extern "C" int ev2h_side_stream_probe(ev2h_stream_t stream, int spin_us, float* ratio) {
    EV2H_CHECK_ARG(ratio && spin_us > 0 && spin_us <= 100000);
    *ratio = 0.f;
    g_side_claim = true;
    SideCtx* side = side_ctx(stream);
    g_side_claim = false;
    if (!side || g_side_disabled) {
        ev2h_set_error("ev2h_side_stream_probe: the side stream is switched off");
        return EV2H_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e[4] = {};
    for (auto& x : e) {
        if (hipEventCreate(&x) != hipSuccess) {
            for (auto& y : e) if (y) (void)hipEventDestroy(y);
            ev2h_set_error("ev2h_side_stream_probe: hipEventCreate failed");
            return EV2H_ERR_HIP;
        }
    }
    int rate_khz = 100000, dev = 0;                         // wall_clock64 ticks per millisecond
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
    const unsigned long long ticks = (unsigned long long)rate_khz * (unsigned long long)spin_us / 1000ull;
    auto cleanup = [&]() { for (auto& x : e) (void)hipEventDestroy(x); };
    float one = 0.f, two = 0.f;
    hipError_t err = hipSuccess;
    for (int rep = 0; rep < 2 && err == hipSuccess; ++rep) {   // (first repetition: code load)
        err = hipEventRecord(e[0], st);
        spin_kernel<<<1, 64, 0, st>>>(ticks);
        if (err == hipSuccess) err = hipEventRecord(e[1], st);
        // the pair: fork exactly as ev2h_forward does
        if (err == hipSuccess) err = hipEventRecord(side->ev[4], st);
        if (err == hipSuccess) err = hipStreamWaitEvent(side->stream, side->ev[4], 0);
        if (err == hipSuccess) err = hipEventRecord(e[2], st);
        spin_kernel<<<1, 64, 0, st>>>(ticks);
        spin_kernel<<<1, 64, 0, side->stream>>>(ticks);
        if (err == hipSuccess) err = hipEventRecord(side->ev[3], side->stream);
        if (err == hipSuccess) err = hipStreamWaitEvent(st, side->ev[3], 0);
        if (err == hipSuccess) err = hipEventRecord(e[3], st);
        if (err == hipSuccess) err = hipStreamSynchronize(st);
    }
    if (err == hipSuccess) err = hipEventElapsedTime(&one, e[0], e[1]);
    if (err == hipSuccess) err = hipEventElapsedTime(&two, e[2], e[3]);
    cleanup();
    if (err != hipSuccess) { ev2h_set_error("ev2h_side_stream_probe: %s", hipGetErrorString(err)); return EV2H_ERR_HIP; }
    *ratio = one > 0.f ? two / one : 0.f;
    return EV2H_OK;
}

// ---------------------------------------------------------------------------------------- shader-clock probe
namespace {
// one wave: shader-clock cycles (s_memtime) and constant-rate reference ticks (s_memrealtime, 100 MHz) over ~ticks reference ticks
__global__ void clock_probe_kernel(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
}  // namespace

extern "C" int ev2h_shader_clock_probe(ev2h_stream_t stream, int spin_us, unsigned long long* out_dev) {
    EV2H_CHECK_ARG(out_dev && spin_us > 0 && spin_us <= 100000);
    clock_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>((unsigned long long)spin_us * 100ull, out_dev);      // s_memrealtime: 100 MHz
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

// ---------------------------------------------------------------------------------------- small kernels
// internal entry points of other translation units (not part of the C ABI)
size_t ev2h_fps_state_ld(int N);
int ev2h_fps_multi_chunk(const float* pts4, int B, int N, int njobs, const int* S, const int64_t* const* init, int32_t* const* idx, float* const* ctr4,
                         int s_begin, int s_end, float* state, ev2h_stream_t stream);
int ev2h_ball_query_range(const float* pts4, const float* ctr4, int B, int N, int S, int s_off, int s_cnt, int nrad, const double* radius,
                          const int* nsample, int32_t* const* gidx, int32_t* cnt, ev2h_stream_t stream);
bool ev2h_gemm_bf16_zsum_supported(const ev2h_gemm_desc* d);
int ev2h_gemm_bf16_zsum(const ev2h_gemm_desc* d, const float* key_pm, float* zpart, int x_bf16, ev2h_stream_t stream, const float* x_scale = nullptr);
int ev2h_fp_mlp_ex(const ev2h_fp_desc* d, int t_bf16, int out_bf16, ev2h_stream_t stream, float* row16_scale = nullptr, float w3_norm = 0.f, float b3_max = 0.f);
int ev2h_attn_context_f16rows(const float* sim, const void* value_pm, int ldv, int B, int N, float* hf8, uint32_t* hf_amax, int amax_hand_stride,
                              const float* value_unscale, const float* vscale, ev2h_stream_t stream);
int ev2h_attn_context_bf16rows(const float* sim, const void* value_pm, int ldv, int B, int N, float* hf8, const float* value_unscale, ev2h_stream_t stream);
int ev2h_attn_simfold_partials(const float* zpart, int rows_per_partial, const float* logits_pm, int B, int N, const float* w4t_left,
                               const float* w4t_right, const float* b4_left, const float* b4_right, float* sim, ev2h_stream_t stream);

namespace {

// ---------------------------------------------------------------------------------------- workspace layout
struct Buf {
    const char* name;
    size_t off;     // bytes
    size_t count;   // elements (4 bytes each)
};

struct Layout {
    static constexpr int MAXB = 96;
    Buf bufs[MAXB];
    int n = 0;
    size_t total = 0;
    size_t add(const char* name, size_t count) {
        const size_t off = total;
        bufs[n++] = Buf{name, off, count};
        total += (count * 4 + 255) / 256 * 256;
        return off;
    }
    const Buf* find(const char* name) const {
        for (int i = 0; i < n; ++i)
            if (!strcmp(bufs[i].name, name)) return &bufs[i];
        return nullptr;
    }
};

// names with an L/R suffix are stored as literals so Buf::name stays valid
static const char* const kHandNames[2][10] = {
    {"P1mL", "fpsmL", "ctrmL", "gidxm0L", "gidxm1L", "cntmL", "m1bufL", "msa2hL", "m2L", "fc1L"},
    {"P1mR", "fpsmR", "ctrmR", "gidxm0R", "gidxm1R", "cntmR", "m1bufR", "msa2hR", "m2R", "fc1R"}};

// F16X2 range records (ev2hands_hip.h "Range records"): one uint32 [B] array per tensor that a contraction reads
enum RangeId {
    R_FEAT, R_L1A, R_L1B, R_L2, R_SA3H1, R_SA3H2, R_L3, R_FP3H, R_FP3O, R_FP2H, R_L1NEW, R_FP1IN, R_FP1H1, R_FP1H2, R_L0, R_CLSH, R_Q1,
    R_HF,            // two records: left, right
    R_HF_R,
    R_M1, R_M1_R, R_MSA2H, R_MSA2H_R, R_M2, R_M2_R, R_FC1, R_FC1_R,
    R_P1A, R_P1B, R_P1M, R_P1M_R, R_FP1T,
    R_COUNT
};
static const char* const kRangeNames[R_COUNT] = {
    "feat", "l1a", "l1b", "l2", "sa3h1", "sa3h2", "l3", "fp3h", "fp3o", "fp2h", "l1new", "fp1in", "fp1h1", "fp1h2", "l0", "clsh", "q1",
    "hfL", "hfR", "m1L", "m1R", "msa2hL", "msa2hR", "m2L", "m2R", "fc1L", "fc1R", "p1a", "p1b", "p1mL", "p1mR", "fp1t"};

static void build_layout(Layout& L, int B, int N) {
    const size_t R = (size_t)B * N;
    const size_t b = (size_t)B;
    L.add("pts4", R * 4);
    L.add("feat8", R * 8);
    L.add("fps1", b * 512);
    L.add("ctr1", b * 512 * 4);
    L.add("P1a", R * 160);
    L.add("gidx1_0", b * 512 * 32);
    L.add("gidx1_1", b * 512 * 64);
    L.add("gidx1_2", b * 512 * 128);
    L.add("cnt1", b * 512 * 3);
    L.add("l1cat", b * 512 * 576);
    L.add("P1b", b * 512 * 256);
    L.add("fps2", b * 128);
    L.add("ctr2", b * 128 * 4);
    L.add("gidx2_0", b * 128 * 64);
    L.add("gidx2_1", b * 128 * 128);
    L.add("cnt2", b * 128 * 2);
    L.add("l2buf", b * 128 * 520);
    L.add("sa3h1", b * 128 * 256);
    L.add("sa3h2", b * 128 * 512);
    L.add("l3", b * 1024);
    L.add("fp3bias", b * 256);
    L.add("fp3h", b * 128 * 256);
    L.add("fp3o", b * 128 * 256);
    L.add("fp2h", b * 512 * 256);
    L.add("l1new", b * 512 * 128);
    L.add("fp1T", b * 512 * 128);                // 16-bit modes: layer-1 table of fp1 (fp1in / fp1h1 / fp1h2 are then unused)
    L.add("fp1in", R * 128);
    L.add("fp1h1", R * 128);
    L.add("fp1h2", R * 128);
    L.add("l0", R * 256);
    L.add("clsh", R * 256);
    L.add("logits_pm", R * 4);
    L.add("q1", R * 512);
    L.add("zpart", std::max(ev2h_attn_sim_folded_scratch(B, N), (size_t)B * ceil_div(N, 128) * 12 * 512));     // (fused form: one partial per 128 rows)
    L.add("sim", b * 2 * 4 * 256);
    L.add("hf8", 2 * R * 8);
    L.add("nn2_idx", b * 512 * 3);
    L.add("nn2_w", b * 512 * 3);
    L.add("nn1_idx", R * 3);
    L.add("nn1_w", R * 3);
    for (int h = 0; h < 2; ++h) {
        L.add(kHandNames[h][0], R * 256);
        L.add(kHandNames[h][1], b * 128);
        L.add(kHandNames[h][2], b * 128 * 4);
        L.add(kHandNames[h][3], b * 128 * 64);
        L.add(kHandNames[h][4], b * 128 * 128);
        L.add(kHandNames[h][5], b * 128 * 2);
        L.add(kHandNames[h][6], b * 128 * 520);
        L.add(kHandNames[h][7], b * 128 * 256);
        L.add(kHandNames[h][8], b * 512);
        L.add(kHandNames[h][9], b * 1024);
    }
    L.add("ranges", (size_t)R_COUNT * b);       // F16X2 range records (uint32 [R_COUNT][B]) ...
    L.add("fps_state", b * 3 * ev2h_fps_state_ld(N));      // chunked sampling of small batches: running minima between the launches
    L.add("p1scale", 6 * b);                    // ... and the storage scales of the five layer-1 tables (float [5][B]) + [5]: of l0 when it is stored as fp16 (F16)
}

struct Ws {
    char* base;
    Layout L;
    int B = 0;
    bool ranges_on = false;      // F16X2: range records are maintained and used
    float* f(const char* name) const { return reinterpret_cast<float*>(base + L.find(name)->off); }
    int32_t* i(const char* name) const { return reinterpret_cast<int32_t*>(base + L.find(name)->off); }
    uint32_t* r(int id) const { return ranges_on ? reinterpret_cast<uint32_t*>(base + L.find("ranges")->off) + (size_t)id * B : nullptr; }
    float* p1scale(int k) const { return ranges_on ? f("p1scale") + (size_t)k * B : nullptr; }
};

// range arguments of one dense layer: where X's record(s) live and where Y's goes
struct Rng {
    const uint32_t* xa = nullptr; const uint32_t* xa2 = nullptr; int xg = 0;
    uint32_t* ya = nullptr; int yg = 0;
};

#define RUN(expr)                 \
    do {                          \
        int rc__ = (expr);        \
        if (rc__) return rc__;    \
    } while (0)

static thread_local int g_precision = EV2H_PREC_F32;   // set by ev2h_forward for the helpers below (single in-flight forward per thread)
static thread_local int g_f16_families = 0;            // F16 mode: the EV2H_FAM_* families on one fp16 plane (ev2h_weights.f16_families)
// precision of one kernel family: the F16 mode runs the families outside its mask as F16X2 (same range records, images packed to match)
static int fam_prec(int prec, int fam) { return prec == EV2H_PREC_F16 && !(g_f16_families & fam) ? EV2H_PREC_F16X2 : prec; }

static int dense(const ev2h_dense& w, const float* X, int ldx, int M, float* Y, int ldy, int relu, ev2h_stream_t st, const Rng& rg,
                 const float* group_bias = nullptr, int group_rows = 0, int ldbias = 0, int taps = 1, int rows_per_seq = 0,
                 int rowmax_rows = 0, int skinny = 0, int fam = EV2H_FAM_DENSE) {
    ev2h_gemm_desc d{};
    d.skinny = skinny;
    d.x_amax = rg.xa; d.x_amax2 = rg.xa2; d.x_group_rows = rg.xg; d.y_amax = rg.ya; d.y_group_rows = rg.yg;
    d.X = X; d.ldx = ldx; d.W = w.W; d.ldw = w.ldw; d.Y = Y; d.ldy = ldy;
    d.M = M; d.N = w.O; d.K = w.K;
    d.bias = group_bias ? group_bias : w.b;
    d.bias_group_rows = group_rows; d.ldbias = ldbias;
    d.relu = relu; d.post_scale = w.post_scale; d.post_shift = w.post_shift;
    d.taps = taps; d.rows_per_seq = rows_per_seq; d.rowmax_rows = rowmax_rows;
    d.precision = fam_prec(g_precision, fam);
    d.Ws = (g_precision != EV2H_PREC_F32) ? w.Ws : nullptr;
    d.ws_tile_rows = w.ws_tile_rows;
    d.w_unscale = w.w_unscale;
    return ev2h_gemm(&d, st);
}

// one multi-scale set abstraction given its selections: layer-1 table GEMM + one fused kernel per radius.
// Range records (F16X2): feat_amax / feat_amax2 = records of the table's input rows, p1_amax / p1_scale = record and storage scale
// of the table, out_amax = record of the module's output.
// chain: the table is read by the fused SET-ABSTRACTION kernels (not by a row chain).  F16 [r6]: those keep one power of two per window
// for the whole chain, so the table's storage scale s must also keep H2' = (s / u2) H2 below 2^15 for every branch:
//   bound' = max(B1, max_br (|W2|_1 B1 + max|b2|) / u2)  <=  [alpha max(1, max_br |W2|_1 / u2)] max|X| + max(beta, max_br (|W2|_1 beta + max|b2|) / u2)
// with B1 = alpha max|X| + beta the layer-1 bound of the other modes (ev2h_sa_desc.p1_scale, F16 contract).
static int sa_table(int precision, const ev2h_sa_module& m, const float* feat, int ldf, int B, int Npts, float* P1, ev2h_stream_t st,
                    const uint32_t* feat_amax, uint32_t* p1_amax, float* p1_scale, bool chain = true) {
    int c1sum = 0;
    float extra = 0.f;                   // max over the branches of |W1x|_1 * radius: what layer 1 adds to a table entry
    for (int i = 0; i < m.nbranch; ++i) {
        c1sum += m.br[i].C1;
        extra = fmaxf(extra, m.br[i].w1x_norm * (float)m.br[i].radius * 1.0000002f);
    }
    ev2h_gemm_desc g{};
    g.X = feat; g.ldx = ldf; g.W = m.W1f; g.ldw = m.kf; g.Y = P1; g.ldy = c1sum;
    g.M = B * Npts; g.N = c1sum; g.K = m.kf; g.bias = m.b1; g.taps = 1;
    g.precision = precision;
    g.w_unscale = m.w1f_unscale;
    if (precision != EV2H_PREC_F32 && m.W1fs) { g.Ws = m.W1fs; g.ws_tile_rows = 128; }
    if (feat_amax) {
        g.x_amax = feat_amax; g.x_group_rows = Npts;
        g.y_amax = p1_amax; g.y_group_rows = Npts;
        g.y_scale = p1_scale; g.y_bound_w = m.w1f_norm; g.y_bound_b = m.b1_max + extra;
        if (precision == EV2H_PREC_F16 && chain) {
            float wmul = 1.f, badd = g.y_bound_b;
            for (int i = 0; i < m.nbranch; ++i) {
                const float iu = 1.000001f / (m.br[i].w2_unscale > 0.f ? m.br[i].w2_unscale : 1.f);
                wmul = fmaxf(wmul, m.br[i].w2_norm * iu);
                badd = fmaxf(badd, (m.br[i].w2_norm * g.y_bound_b + m.br[i].b2_max) * iu);
            }
            g.y_bound_w *= wmul; g.y_bound_b = badd;
        }
    }
    return ev2h_gemm(&g, st);
}

// Plane modes with raw feature rows (kf == 8: enc.sa1, the regressors' sa1): layer 1 runs on the matrix pipe inside the fused kernel
// straight from the feature rows (ev2h_sa_desc.feat) -- no layer-1 table is computed, written (1.46 GB per 256-window step) or
// gathered.  (BF16, F16X2: round 4; BF16X3: round 5.)  EV2H_L1_TABLE=1: A/B switch back to the table (the path F32 always takes).
static bool bf16_direct_layer1(int precision, const ev2h_sa_module& m) {
    static const bool table = getenv("EV2H_L1_TABLE") != nullptr;
    return precision != EV2H_PREC_F32 && m.kf == 8 && !table;
}

static int sa_branches(int precision, const char* tag, const ev2h_sa_module& m, const float* pts4, const float* ctr4, int32_t* const* gidx,
                       const int32_t* cnt, int B, int Npts, const float* P1, float* out, int ldo, ev2h_stream_t st, bool ranges,
                       const uint32_t* p1_amax, const float* p1_scale, uint32_t* out_amax, const float* feat = nullptr, int nfeat = 0,
                       const uint32_t* feat_amax = nullptr, float* xyz_out = nullptr, int xyz_ld = 0, int s_off = 0, int s_cnt = 0) {
    int c1sum = 0;
    for (int i = 0; i < m.nbranch; ++i) c1sum += m.br[i].C1;
    int coff1 = 0, coff3 = 0;
    for (int i = 0; i < m.nbranch; ++i) {
        const ev2h_sa_branch& br = m.br[i];
        ev2h_sa_desc d{};
        d.P1 = P1 + coff1; d.ldp = c1sum; d.pts4 = pts4; d.ctr4 = ctr4; d.gidx = gidx[i];
        d.W1x = br.W1x; d.W2 = br.W2; d.b2 = br.b2; d.W3 = br.W3; d.b3 = br.b3;
        d.out = out + coff3; d.ldo = ldo;
        d.B = B; d.Npts = Npts; d.S = m.npoint; d.K = br.K; d.C1 = br.C1; d.C2 = br.C2; d.C3 = br.C3;
        if (s_cnt > 0) { d.S = s_cnt; d.S_total = m.npoint; d.s_off = s_off; }      // the centroids [s_off, s_off + s_cnt) of every window
        d.precision = precision; d.W2s = br.W2s; d.W3s = br.W3s; d.w2_unscale = br.w2_unscale; d.w3_unscale = br.w3_unscale;
        const bool direct = feat && bf16_direct_layer1(precision, m);
        if (direct) {
            d.feat = feat; d.ldf = 8; d.W1f = m.W1f + (size_t)coff1 * m.kf; d.ldw1f = m.kf; d.b1 = m.b1 + coff1; d.nfeat = nfeat;
            d.w1f_unscale = br.w1f_unscale; d.w1x_unscale = br.w1x_unscale;
            if (ranges) { d.feat_amax = feat_amax; d.w1f_norm = m.w1f_norm; d.b1_max = m.b1_max; }
        }
        if (ranges) {
            if (!direct) { d.p1_scale = p1_scale; d.p1_amax = p1_amax; }
            d.out_amax = out_amax;
            d.w1x_norm = br.w1x_norm; d.dmax = (float)br.radius * 1.0000002f /* rounded up: a bound */; d.w2_norm = br.w2_norm; d.b2_max = br.b2_max;
        }
        d.cnt = cnt ? cnt + i : nullptr; d.cnt_ld = m.nbranch;      // padding-only strips are skipped (bit-identical: test_sa_mlp_max_skips_padding_strips)
        if (i == 0 && xyz_out) { d.xyz_out = xyz_out; d.xyz_ld = xyz_ld; }      // the consumer's raw-xyz columns: written once, by the first branch
        char t[40];
        snprintf(t, sizeof(t), "%s.%d", tag, i);
        prof_begin(t, st);
        RUN(ev2h_sa_mlp_max(&d, st));
        prof_end(t, st);
        coff1 += br.C1;
        coff3 += br.C3;
    }
    return EV2H_OK;
}

static int sa_module(int precision, const char* tag, const ev2h_sa_module& m, const float* feat, int ldf, const float* pts4, const float* ctr4,
                     int32_t* const* gidx, const int32_t* cnt, int B, int Npts, float* P1, float* out, int ldo, ev2h_stream_t st,
                     const uint32_t* feat_amax, uint32_t* p1_amax, float* p1_scale, uint32_t* out_amax, int nfeat = 0, float* xyz_out = nullptr,
                     int xyz_ld = 0) {
    if (!bf16_direct_layer1(precision, m)) RUN(sa_table(precision, m, feat, ldf, B, Npts, P1, st, feat_amax, p1_amax, p1_scale));
    return sa_branches(precision, tag, m, pts4, ctr4, gidx, cnt, B, Npts, P1, out, ldo, st, feat_amax != nullptr, p1_amax, p1_scale, out_amax,
                       ldf == 8 ? feat : nullptr, nfeat, feat_amax, xyz_out, xyz_ld);
}

}  // namespace

// ---------------------------------------------------------------------------------------- F16X2 spread report
// The operand tensors of the F16X2 contractions that are MATERIALISED in the workspace, as their consumers read them: buffer,
// rows per window, row stride, column range, and the range record(s) the consumer derives its power-of-two scale from (two
// records: the consumer takes their maximum -- a concatenated input).  Not listed: operands that never reach memory (the hidden
// layers inside the fused set-abstraction / row-chain kernels, whose scales come from bounds): TEHNet.verify_precision compares
// whole forwards for those.
static thread_local int g_last_l0_bf16 = 0;      // how the calling thread's last ev2h_forward stored l0: 0 float32, 1 bf16 (BF16), 2 fp16 x p1scale[5][b] (F16)

struct SpreadEntry { const char* name; const char* buf; int rows; int ld; int col0; int ncols; int rec; int rec2; size_t hand_off; };

static int spread_entries(int N, SpreadEntry* e) {      // rows == 0: N rows per window
    int n = 0;
    e[n++] = {"feat", "feat8", 0, 8, 0, 8, R_FEAT, -1, 0};
    e[n++] = {"l1", "l1cat", 512, 576, 0, 320, R_L1A, -1, 0};
    e[n++] = {"l1cat", "l1cat", 512, 576, 0, 576, R_L1A, R_L1B, 0};
    e[n++] = {"l2", "l2buf", 128, 520, 0, 515, R_L2, R_FEAT, 0};
    e[n++] = {"sa3h1", "sa3h1", 128, 256, 0, 256, R_SA3H1, -1, 0};
    e[n++] = {"sa3h2", "sa3h2", 128, 512, 0, 512, R_SA3H2, -1, 0};
    e[n++] = {"l3", "l3", 1, 1024, 0, 1024, R_L3, -1, 0};
    e[n++] = {"fp3h", "fp3h", 128, 256, 0, 256, R_FP3H, -1, 0};
    e[n++] = {"fp3o", "fp3o", 128, 256, 0, 256, R_FP3O, -1, 0};
    e[n++] = {"fp2h", "fp2h", 512, 256, 0, 256, R_FP2H, -1, 0};
    e[n++] = {"l1new", "l1new", 512, 128, 0, 128, R_L1NEW, -1, 0};
    e[n++] = {"l0", "l0", 0, 256, 0, 256, R_L0, -1, 0};
    for (int h = 0; h < 2; ++h) {
        e[n++] = {h ? "hfR" : "hfL", "hf8", 0, 8, 0, 8, R_HF + h, -1, (size_t)h};
        e[n++] = {h ? "m1R" : "m1L", kHandNames[h][6], 128, 520, 0, 515, R_M1 + h, R_FEAT, 0};
        e[n++] = {h ? "msa2hR" : "msa2hL", kHandNames[h][7], 128, 256, 0, 256, R_MSA2H + h, -1, 0};
        e[n++] = {h ? "m2R" : "m2L", kHandNames[h][8], 1, 512, 0, 512, R_M2 + h, -1, 0};
        e[n++] = {h ? "fc1R" : "fc1L", kHandNames[h][9], 1, 1024, 0, 1024, R_FC1 + h, -1, 0};
    }
    (void)N;
    return n;
}
constexpr int EV2H_MAX_SPREAD = 32;

namespace {
// counts[b] = {non-zero values, values with 0 < |v| s < 2^-3 (low fp16 plane subnormal: fewer than 22 bits survive the split),
// values with 0 < |v| s < 2^-14 (high plane subnormal too: fewer than 11 bits)}, s = the consumer's power-of-two scale
// half_elems: the buffer holds fp16 values (F16 mode's l0: stored times a power of two, and so is its record -- the ratios are the same)
__global__ __launch_bounds__(256) void spread_count_kernel(const float* __restrict__ buf, size_t window_stride, int rows, int ld, int col0, int ncols,
                                                           const unsigned* __restrict__ rec, const unsigned* __restrict__ rec2,
                                                           unsigned* __restrict__ counts, int half_elems) {
    const int b = blockIdx.y;
    unsigned a = rec[b];
    if (rec2) a = max(a, rec2[b]);
    const float s = f16x2_scale(a);
    const float* base = buf + (size_t)b * window_stride;
    const size_t total = (size_t)rows * ncols;
    unsigned nz = 0, lo = 0, hi = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / ncols;
        const int c = (int)(i - r * ncols);
        const float v = fabsf(half_elems ? (float)reinterpret_cast<const _Float16*>(buf)[(size_t)b * window_stride + r * ld + col0 + c] : base[r * ld + col0 + c]) * s;
        nz += v > 0.f;
        lo += v > 0.f && v < 0.125f;
        hi += v > 0.f && v < 6.103515625e-05f;
    }
    nz = (unsigned)wave_sum_f32((float)nz); lo = (unsigned)wave_sum_f32((float)lo); hi = (unsigned)wave_sum_f32((float)hi);   // < 2^24 per wave: exact
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[(size_t)b * 3 + 0], nz);
        atomicAdd(&counts[(size_t)b * 3 + 1], lo);
        atomicAdd(&counts[(size_t)b * 3 + 2], hi);
    }
}
}  // namespace

extern "C" int ev2h_range_report_entries(const char** names, int max_names) {
    SpreadEntry e[EV2H_MAX_SPREAD];
    const int n = spread_entries(2048, e);
    for (int i = 0; i < n && names && i < max_names; ++i) names[i] = e[i].name;
    return n;
}

extern "C" int ev2h_range_report(void* workspace, int B, int N, uint32_t* counts, ev2h_stream_t st) {
    EV2H_CHECK_ARG(workspace && counts && B > 0 && N >= 128 && N <= 32768);
    Ws ws;
    ws.base = static_cast<char*>(workspace);
    ws.B = B;
    ws.ranges_on = true;
    build_layout(ws.L, B, N);
    SpreadEntry e[EV2H_MAX_SPREAD];
    const int n = spread_entries(N, e);
    EV2H_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)n * B * 3 * sizeof(uint32_t), (hipStream_t)st));
    for (int i = 0; i < n; ++i) {
        const Buf* bf = ws.L.find(e[i].buf);
        EV2H_CHECK_ARG(bf != nullptr);
        const int rows = e[i].rows ? e[i].rows : N;
        const float* p = reinterpret_cast<const float*>(ws.base + bf->off) + e[i].hand_off * (size_t)B * N * 8;      // hf8: [2][B * N][8]
        const size_t per_window = (size_t)rows * e[i].ld;
        const int gx = (int)std::min<size_t>(64, ((size_t)rows * e[i].ncols + 4095) / 4096);
        spread_count_kernel<<<dim3(std::max(gx, 1), B), 256, 0, (hipStream_t)st>>>(p, per_window, rows, e[i].ld, e[i].col0, e[i].ncols, ws.r(e[i].rec),
                                                                                  e[i].rec2 >= 0 ? ws.r(e[i].rec2) : nullptr, counts + (size_t)i * B * 3,
                                                                                  (!strcmp(e[i].buf, "l0") && g_last_l0_bf16 == 2) ? 1 : 0);
        EV2H_CHECK_LAUNCH();
    }
    return EV2H_OK;
}

extern "C" size_t ev2h_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    Layout L;
    build_layout(L, B, N);
    return L.total;
}

extern "C" const void* ev2h_workspace_buffer_ex(void* workspace, int B, int N, const char* name, size_t* count, int* elem_type) {
    const void* p = ev2h_workspace_buffer(workspace, B, N, name, count);
    if (elem_type) *elem_type = (p && !strcmp(name, "l0")) ? g_last_l0_bf16 : 0;      // 1 = bf16, 2 = fp16 times p1scale[5][b]
    return p;
}

extern "C" const void* ev2h_workspace_buffer(void* workspace, int B, int N, const char* name, size_t* count) {
    if (!workspace || !name || B <= 0 || N <= 0) return nullptr;
    Layout L;
    build_layout(L, B, N);
    if (!strncmp(name, "rng.", 4)) {                  // one F16X2 range record: "rng.<tensor>" -> uint32 [B]
        for (int i = 0; i < R_COUNT; ++i)
            if (!strcmp(name + 4, kRangeNames[i])) {
                if (count) *count = (size_t)B;
                return static_cast<char*>(workspace) + L.find("ranges")->off + (size_t)i * B * 4;
            }
        return nullptr;
    }
    const Buf* b = L.find(name);
    if (!b) return nullptr;
    if (count) *count = b->count;
    return static_cast<char*>(workspace) + b->off;
}

static int forward_body(const ev2h_weights* w, const ev2h_mano_consts* const* mano, float* xyz_cm, int B, int C, int N, int mhlnes,
                        const int64_t* fps_init, const ev2h_outputs* out, const Ws& ws, ev2h_stream_t st, SideCtx* side, bool* forked) {
    const int R = B * N;
    const int prec = w->precision;
    // F16: the precision each kernel family runs in (ev2h_weights.f16_families; everything else: prec itself)
    const int prec_sa = fam_prec(prec, EV2H_FAM_SA), prec_rows = fam_prec(prec, EV2H_FAM_ROWS), prec_q = fam_prec(prec, EV2H_FAM_QCONV);
    auto rg = [&](int xid, int xg, int yid = -1, int yg = 0, int xid2 = -1) {
        Rng r{};
        if (ws.ranges_on) {
            r.xa = ws.r(xid); r.xg = xg;
            if (xid2 >= 0) r.xa2 = ws.r(xid2);
            if (yid >= 0) { r.ya = ws.r(yid); r.yg = yg; }
        }
        return r;
    };
    if (ws.ranges_on) {
        const Buf* rb = ws.L.find("ranges");
        EV2H_CHECK_HIP(hipMemsetAsync(ws.base + rb->off, 0, rb->count * 4, (hipStream_t)st));
    }

    // ---- input layout + all three samplings of the raw cloud (enc.sa1, left.sa1, right.sa1)
    RUN(ev2h_prep_points(xyz_cm, B, C, N, mhlnes, ws.f("pts4"), ws.f("feat8"), ws.r(R_FEAT), st));
    // fork 0: the layer-1 table of enc.sa1 needs the prepared input only; it is written (HBM-bound) on the side stream while the
    // farthest-point sampling (latency-bound, 896 dependent steps) and the ball query run on the caller's stream
    const bool fork = side != nullptr;
    // sd: the side stream (or the caller's in single-stream mode, EV2H_TWO_STREAMS=0 / ev2h_set_side_stream(0)).  On it: the
    // layer-1 table, every selection that needs only coordinates, the classifier, and the right hand's regressor.
    ev2h_stream_t sd = fork ? (ev2h_stream_t)side->stream : st;
    ev2h_stream_t sx = sd;
    ev2h_stream_t sc = sd;
    if (fork) {
        EV2H_CHECK_HIP(hipEventRecord(side->ev[4], (hipStream_t)st));
        EV2H_CHECK_HIP(hipStreamWaitEvent(side->stream, side->ev[4], 0));
        *forked = true;
    }
    if (!bf16_direct_layer1(prec, w->sa1)) RUN(sa_table(prec_sa, w->sa1, ws.f("feat8"), 8, B, N, ws.f("P1a"), sx, ws.r(R_FEAT), ws.r(R_P1A), ws.p1scale(0)));
    if (fork) EV2H_CHECK_HIP(hipEventRecord(side->ev[5], side->stream));
    // [r6] enc.sa1's sampling is 512 DEPENDENT arg-max steps on 3 workgroups per window -- at 16 windows of 8192 points 0.49 ms on 48
    // of 256 CUs (a fifth of the forward), at 8 windows of 2048 points 0.25 of 1.25 ms -- and everything else waited for it.  The 512
    // centroids are now drawn in four launches of 128 on the SIDE stream (ev2h_fps_multi_chunk: the running minima travel through
    // the workspace; the same arg-max sequence, identical indices), and the caller's stream runs the ball query and the three fused
    // set-abstraction launches of each quarter as soon as it is drawn (ev2h_sa_desc.s_off): 3/4 of enc.sa1 runs beside the sampling.
    // Same kernels on the same groups: bit-identical outputs.  Measured (tools/debug/chunk_check.py, profiles/r6_fps_chunks*.txt):
    // 16 x 8192: 6 969 -> 7 612 windows/s, 8 x 2048 hipGraph 1.177 -> 1.113 ms, 32 x 2048 +4 %, 64 x 8192 +2-3 %, 256 x 2048 +1.0 %;
    // NOT below 8 windows (the twelve extra launches cost more than the overlap: -1.5 .. -3 %) and not where the sampling already
    // fills the chip with one staged window per CU (128 x 8192: -1.5 %).
    // EV2H_FPS_CHUNKS=0: A/B switch (one launch); = n > 1: chunk at any batch up to n sampling workgroups (tuning).
    static const int chunk_env = [] { const char* e = getenv("EV2H_FPS_CHUNKS"); return e ? atoi(e) : -1; }();
    constexpr int FPS_CHUNKS = 4;
    const int chunk_max_wg = chunk_env > 1 ? chunk_env : (N <= 2048 ? 0x7fffffff : 256), chunk_min_b = chunk_env > 1 ? 1 : 8;
    const bool chunked = fork && chunk_env != 0 && B >= chunk_min_b && 3 * B <= chunk_max_wg && bf16_direct_layer1(prec, w->sa1) && w->sa1.npoint % FPS_CHUNKS == 0;
    {
        const int S[3] = {512, 128, 128};
        const int64_t* init[3] = {fps_init, fps_init + 2 * (size_t)B, fps_init + 3 * (size_t)B};
        int32_t* idx[3] = {ws.i("fps1"), ws.i("fpsmL"), ws.i("fpsmR")};
        float* ctr[3] = {ws.f("ctr1"), ws.f("ctrmL"), ws.f("ctrmR")};
        if (!chunked) {
            RUN(ev2h_fps_multi(ws.f("pts4"), B, N, 3, S, init, idx, ctr, st));
        } else {
            const int q = w->sa1.npoint / FPS_CHUNKS;
            for (int c = 0; c < FPS_CHUNKS; ++c) {          // (the regressors' 128-centroid samplings finish inside the first launch)
                RUN(ev2h_fps_multi_chunk(ws.f("pts4"), B, N, 3, S, init, idx, ctr, c * q, (c + 1) * q, ws.f("fps_state"), sd));
                EV2H_CHECK_HIP(hipEventRecord(side->ev[10 + c], side->stream));
            }
        }
    }
    // fork 1: everything that needs only COORDINATES runs on the side stream, under the MFMA-bound set-abstraction kernels of
    // enc.sa1 on the caller's stream (it used to sit between them on the critical path): the sampling and the ball query of
    // enc.sa2 (its input points are enc.sa1's centroids), the 3-NN selection of fp1 (raw cloud against those centroids), then
    // both hands' ball queries.  Same kernels, same inputs: bit-identical.
    if (fork && !chunked) {                                   // (chunked: the sampling itself ran on the side stream)
        EV2H_CHECK_HIP(hipEventRecord(side->ev[0], (hipStream_t)st));
        EV2H_CHECK_HIP(hipStreamWaitEvent(side->stream, side->ev[0], 0));
    }
    int32_t* gi2[2] = {ws.i("gidx2_0"), ws.i("gidx2_1")};
    {
        const ev2h_sa_module& m = w->sa2;
        RUN(ev2h_fps(ws.f("ctr1"), B, 512, 128, fps_init + (size_t)B, ws.i("fps2"), ws.f("ctr2"), sc));
        double rad[2]; int ns[2];
        for (int i = 0; i < 2; ++i) { rad[i] = m.br[i].radius; ns[i] = m.br[i].K; }
        RUN(ev2h_ball_query(ws.f("ctr1"), ws.f("ctr2"), B, 512, 128, 2, rad, ns, gi2, ws.i("cnt2"), sc));
        if (fork) EV2H_CHECK_HIP(hipEventRecord(side->ev[8], side->stream));
    }
    static const bool unfused_fp1 = getenv("EV2H_FP1_UNFUSED") != nullptr;      // A/B switch
    const bool fp1_fused = prec != EV2H_PREC_F32 && w->fp1m.W1fs && !unfused_fp1;
    if (fp1_fused) {
        RUN(ev2h_three_nn_interp(ws.f("pts4"), ws.f("ctr1"), B, N, 512, nullptr, 0, 0, nullptr, 0, ws.i("nn1_idx"), ws.f("nn1_w"), nullptr, sc));
        if (fork) EV2H_CHECK_HIP(hipEventRecord(side->ev[9], side->stream));
    }
    for (int h = 0; h < 2; ++h) {
        const ev2h_sa_module& m = w->mano_sa1[h];
        const char* const* nm = kHandNames[h];
        double rad[2]; int ns[2];
        int32_t* gi[2] = {ws.i(nm[3]), ws.i(nm[4])};
        for (int i = 0; i < 2; ++i) { rad[i] = m.br[i].radius; ns[i] = m.br[i].K; }
        RUN(ev2h_ball_query(ws.f("pts4"), ws.f(nm[2]), B, N, 128, 2, rad, ns, gi, ws.i(nm[5]), sd));
    }
    if (fork) EV2H_CHECK_HIP(hipEventRecord(side->ev[1], side->stream));
    // ---- enc.sa1 (TEHNet.py:179)
    {
        const ev2h_sa_module& m = w->sa1;
        double rad[3]; int ns[3];
        int32_t* gi[3] = {ws.i("gidx1_0"), ws.i("gidx1_1"), ws.i("gidx1_2")};
        for (int i = 0; i < 3; ++i) { rad[i] = m.br[i].radius; ns[i] = m.br[i].K; }
        if (!chunked) {
            RUN(ev2h_ball_query(ws.f("pts4"), ws.f("ctr1"), B, N, 512, 3, rad, ns, gi, ws.i("cnt1"), st));
            if (fork) EV2H_CHECK_HIP(hipStreamWaitEvent((hipStream_t)st, side->ev[5], 0));          // the table is written
            RUN(sa_branches(prec_sa, "sa1", m, ws.f("pts4"), ws.f("ctr1"), gi, ws.i("cnt1"), B, N, ws.f("P1a"), ws.f("l1cat"), 576, st, ws.ranges_on,
                            ws.r(R_P1A), ws.p1scale(0), ws.r(R_L1A), ws.f("feat8"), C, ws.r(R_FEAT)));
        } else {
            const int q = m.npoint / FPS_CHUNKS;
            for (int c = 0; c < FPS_CHUNKS; ++c) {          // quarter c: as soon as its centroids are drawn
                EV2H_CHECK_HIP(hipStreamWaitEvent((hipStream_t)st, side->ev[10 + c], 0));
                RUN(ev2h_ball_query_range(ws.f("pts4"), ws.f("ctr1"), B, N, 512, c * q, q, 3, rad, ns, gi, ws.i("cnt1"), st));
                RUN(sa_branches(prec_sa, "sa1", m, ws.f("pts4"), ws.f("ctr1"), gi, ws.i("cnt1"), B, N, ws.f("P1a"), ws.f("l1cat"), 576, st, ws.ranges_on,
                                ws.r(R_P1A), ws.p1scale(0), ws.r(R_L1A), ws.f("feat8"), C, ws.r(R_FEAT), nullptr, 0, c * q, q));
            }
        }
    }
    // ---- enc.sa2 (TEHNet.py:180) on the 512 sampled points (sampling + ball query: fork 1 above)
    {
        const ev2h_sa_module& m = w->sa2;
        if (fork) EV2H_CHECK_HIP(hipStreamWaitEvent((hipStream_t)st, side->ev[8], 0));
        RUN(sa_module(prec_sa, "sa2", m, ws.f("l1cat"), 576, ws.f("ctr1"), ws.f("ctr2"), gi2, ws.i("cnt2"), B, 512, ws.f("P1b"), ws.f("l2buf"), 520, st,
                      ws.r(R_L1A), ws.r(R_P1B), ws.p1scale(1), ws.r(R_L2), 0, ws.f("l2buf") + 512, 520));      // + the xyz columns of enc.sa3's input
    }
    // ---- enc.sa3 group-all (TEHNet.py:181): 515 -> 256 -> 512 -> 1024, max over the 128 points
    // (the xyz columns of l2buf are input coordinates: the input's record R_FEAT bounds them)
    RUN(dense(w->sa3[0], ws.f("l2buf"), 520, B * 128, ws.f("sa3h1"), 256, 1, st, rg(R_L2, 128, R_SA3H1, 128, R_FEAT)));
    RUN(dense(w->sa3[1], ws.f("sa3h1"), 256, B * 128, ws.f("sa3h2"), 512, 1, st, rg(R_SA3H1, 128, R_SA3H2, 128)));
    RUN(dense(w->sa3[2], ws.f("sa3h2"), 512, B * 128, ws.f("l3"), 1024, 1, st, rg(R_SA3H2, 128, R_L3, 1), nullptr, 0, 0, 1, 0, 128));
    // ---- fp3 (TEHNet.py:184): the single l3 point is broadcast, so its 1024 inputs collapse to a per-window bias
    RUN(dense(w->fp3_bcast, ws.f("l3"), 1024, B, ws.f("fp3bias"), 256, 0, st, rg(R_L3, 1), nullptr, 0, 0, 1, 0, 0, 1));
    RUN(dense(w->fp3_skip, ws.f("l2buf"), 520, B * 128, ws.f("fp3h"), 256, 1, st, rg(R_L2, 128, R_FP3H, 128, R_FEAT), ws.f("fp3bias"), 128, 256));
    RUN(dense(w->fp3_1, ws.f("fp3h"), 256, B * 128, ws.f("fp3o"), 256, 1, st, rg(R_FP3H, 128, R_FP3O, 128)));
    // ---- fp2 (TEHNet.py:185): 3-NN 128 -> 512, concat [skip 320 | interpolated 256]
    RUN(ev2h_three_nn_interp(ws.f("ctr1"), ws.f("ctr2"), B, 512, 128, ws.f("fp3o"), 256, 256, ws.f("l1cat") + 320, 576,
                             ws.i("nn2_idx"), ws.f("nn2_w"), ws.r(R_L1B), st));
    RUN(dense(w->fp2[0], ws.f("l1cat"), 576, B * 512, ws.f("fp2h"), 256, 1, st, rg(R_L1A, 512, R_FP2H, 512, R_L1B)));
    RUN(dense(w->fp2[1], ws.f("fp2h"), 256, B * 512, ws.f("l1new"), 128, 1, st, rg(R_FP2H, 512, R_L1NEW, 512)));
    // ---- fp1 (TEHNet.py:186): 3-NN 512 -> N, no skip
    // [r5] BF16: l0 -- the forward's one N-row, 256-wide tensor: written once (fp1), read three times (segmentation head, k = 3 query
    // convolution, attention context) -- is stored as bf16 when all four run in their fused forms.  Every BF16 reader rounds it to
    // bf16 before multiplying anyway (the context read it in fp32: it now sees the rounded values, inside the mode's own error).
    // 2.1 of the BF16 step's 6.0 GB of HBM traffic touch l0.  EV2H_L0_F32=1: A/B switch (fp32 l0 in BF16 as well).
    static const bool unfused_cls = getenv("EV2H_CLS_UNFUSED") != nullptr;      // A/B switch
    static const bool unfused_zsum = getenv("EV2H_ATTN_UNFUSED_ZSUM") != nullptr;
    static const bool l0_f32 = getenv("EV2H_L0_F32") != nullptr;
    const bool cls_fused = prec != EV2H_PREC_F32 && w->clsm.W2s && !unfused_cls;
    ev2h_gemm_desc qd{};                     // the first query convolution (both hands in one GEMM), used further down
    {
        const Rng r0 = rg(R_L0, N);
        qd.x_amax = r0.xa; qd.x_amax2 = r0.xa2; qd.x_group_rows = r0.xg;
        qd.X = ws.f("l0"); qd.ldx = 256; qd.W = w->qconv0.W; qd.ldw = w->qconv0.ldw;
        qd.M = R; qd.N = w->qconv0.O; qd.K = w->qconv0.K;
        qd.bias = w->qconv0.b; qd.relu = 1; qd.post_scale = w->qconv0.post_scale; qd.post_shift = w->qconv0.post_shift;
        qd.taps = 3; qd.rows_per_seq = N; qd.precision = prec_q; qd.Ws = w->qconv0.Ws; qd.ws_tile_rows = w->qconv0.ws_tile_rows;
        qd.w_unscale = w->qconv0.w_unscale;
    }
    const bool zsum_ok = !unfused_zsum && prec != EV2H_PREC_F32 && w->qconv0.Ws && ev2h_gemm_bf16_zsum_supported(&qd);
    const bool l0_bf16 = prec == EV2H_PREC_BF16 && fp1_fused && cls_fused && zsum_ok && !l0_f32;
    // [r6] F16: the same tensor as fp16 times a per-window power of two (ws.p1scale(5)): the fp1 chain chooses it from the bound of its own
    // output, the three readers take the stored values as their operand plane (ev2h_fp_mlp_ex, ev2h_gemm_bf16_zsum, ev2h_attn_context_f16rows).
    // Needs the row chains and the query convolution to run one-plane fp16 (ev2h_weights.f16_families) in their fused forms.
    const bool l0_f16 = prec == EV2H_PREC_F16 && prec_rows == EV2H_PREC_F16 && prec_q == EV2H_PREC_F16 && ws.ranges_on && fp1_fused && cls_fused && zsum_ok && !l0_f32;
    const bool l0_16 = l0_bf16 || l0_f16;
    g_last_l0_bf16 = l0_bf16 ? 1 : (l0_f16 ? 2 : 0);       // (ev2h_workspace_buffer_ex tells a debugger what "l0" holds)
    if (fp1_fused) {
        // 16-bit modes: the first layer commutes with the interpolation -- a 512-row table per window instead of an N-row GEMM --
        // and the blend of three table rows, layers 2-3 and the ReLUs run in one kernel (ev2h_fp_mlp): the interpolated rows and
        // the two hidden layers (3 x 268 MB written and read back at B = 256) never reach memory
        const ev2h_sa_module& m = w->fp1m;
        if (fork) EV2H_CHECK_HIP(hipStreamWaitEvent((hipStream_t)st, side->ev[9], 0));      // 3-NN selection: fork 1 above
        RUN(sa_table(prec_rows, m, ws.f("l1new"), 128, B, 512, ws.f("fp1T"), st, ws.r(R_L1NEW), ws.r(R_FP1T), ws.p1scale(4), false));
        ev2h_fp_desc d{};
        d.T = ws.f("fp1T"); d.ldt = 128; d.nn_idx = ws.i("nn1_idx"); d.nn_w = ws.f("nn1_w");
        d.b2 = m.br[0].b2; d.b3 = m.br[0].b3; d.W2s = m.br[0].W2s; d.W3s = m.br[0].W3s;
        d.w2_unscale = m.br[0].w2_unscale; d.w3_unscale = m.br[0].w3_unscale;
        d.out = ws.f("l0"); d.ldo = 256; d.B = B; d.N = N; d.S = 512; d.C1 = 128; d.C2 = 128; d.C3 = 256; d.precision = prec_rows;
        if (ws.ranges_on) {
            d.t_scale = ws.p1scale(4); d.t_amax = ws.r(R_FP1T); d.w2_norm = m.br[0].w2_norm; d.b2_max = m.br[0].b2_max; d.out_amax = ws.r(R_L0);
        }
        prof_begin("fp1", st);
        RUN(ev2h_fp_mlp_ex(&d, 0, l0_16, st, l0_f16 ? ws.p1scale(5) : nullptr, m.br[0].w3_norm, m.br[0].b3_max));
        prof_end("fp1", st);
    } else {
        RUN(ev2h_three_nn_interp(ws.f("pts4"), ws.f("ctr1"), B, N, 512, ws.f("l1new"), 128, 128, ws.f("fp1in"), 128,
                                 ws.i("nn1_idx"), ws.f("nn1_w"), ws.r(R_FP1IN), st));
        RUN(dense(w->fp1[0], ws.f("fp1in"), 128, R, ws.f("fp1h1"), 128, 1, st, rg(R_FP1IN, N, R_FP1H1, N)));      // (un-fused A/B forms: packed as FAM_DENSE)
        RUN(dense(w->fp1[1], ws.f("fp1h1"), 128, R, ws.f("fp1h2"), 128, 1, st, rg(R_FP1H1, N, R_FP1H2, N)));
        RUN(dense(w->fp1[2], ws.f("fp1h2"), 128, R, ws.f("l0"), 256, 1, st, rg(R_FP1H2, N, R_L0, N)));
    }
    // ---- classifier (TEHNet.py:188): independent of the query convolutions (both read l0) -- on the side stream, so that its
    // HBM-bound tail (the 4-column layer, the logits transpose) runs under the k=3 GEMMs
    if (fork) {
        EV2H_CHECK_HIP(hipEventRecord(side->ev[6], (hipStream_t)st));
        EV2H_CHECK_HIP(hipStreamWaitEvent(side->stream, side->ev[6], 0));
    }
    if (cls_fused) {
        // 16-bit modes: both layers in one row-chain kernel -- the 256-wide hidden layer (537 MB at B = 256) never reaches memory,
        // and the logits are written point-major (for the attention) and channel-major (the output) by the same kernel
        const ev2h_sa_branch& c = w->clsm;
        ev2h_fp_desc d{};
        d.T = ws.f("l0"); d.ldt = 256; d.b2 = c.b2; d.b3 = c.b3; d.W2s = c.W2s; d.W3s = c.W3s;
        d.w2_unscale = c.w2_unscale; d.w3_unscale = c.w3_unscale;
        d.out = ws.f("logits_pm"); d.ldo = 4; d.out_cols = 4; d.no_relu_out = 1; d.out_cm = out->class_logits; d.out_cm_stride = out->logits_stride;
        d.B = B; d.N = N; d.C1 = c.C1; d.C2 = c.C2; d.C3 = c.C3; d.precision = prec_rows;
        if (ws.ranges_on) { d.t_amax = ws.r(R_L0); d.w2_norm = c.w2_norm; d.b2_max = c.b2_max; }
        RUN(ev2h_fp_mlp_ex(&d, l0_16, 0, sx, l0_f16 ? ws.p1scale(5) : nullptr));
    } else {
        RUN(dense(w->cls0, ws.f("l0"), 256, R, ws.f("clsh"), 256, 1, sx, rg(R_L0, N, R_CLSH, N)));
        RUN(dense(w->cls4, ws.f("clsh"), 256, R, ws.f("logits_pm"), 4, 0, sx, rg(R_CLSH, N)));
        RUN(ev2h_transpose_logits(ws.f("logits_pm"), B, N, out->class_logits, out->logits_stride, sx));
    }
    if (fork) EV2H_CHECK_HIP(hipEventRecord(side->ev[7], side->stream));
    // ---- query convolutions (TEHNet.py:191-192), both hands' first conv in one GEMM
    // (q1's range record is only needed by the unfolded second convolution: the folded form reads q1 in fp32)
    // [r4; BF16X3: r5] every plane mode: q1 is NOT WRITTEN -- the GEMM's epilogue forms the attention's key-weighted sums of its own tile
    // (gemm_bf16.hip: zsum_epilogue), which makes the logits its input: the classifier is waited for first.
    // EV2H_ATTN_UNFUSED_ZSUM=1: A/B switch (q1 to memory, attn_zsum_kernel reads it back).
    bool zsum_fused = false;
    if (zsum_ok) {
        // the shape preconditions (N % 128 == 0, ...) were tested above (zsum_ok), BEFORE the launch site is bracketed: one event pair
        // per step, and a genuine error of the fused launch is returned, never turned into the two-pass schedule
        if (fork) EV2H_CHECK_HIP(hipStreamWaitEvent((hipStream_t)st, side->ev[7], 0));          // logits ready
        prof_begin("qconv0", st);
        RUN(ev2h_gemm_bf16_zsum(&qd, ws.f("logits_pm"), ws.f("zpart"), l0_16, st, l0_f16 ? ws.p1scale(5) : nullptr));
        prof_end("qconv0", st);
        zsum_fused = true;
    }
    if (!zsum_fused) {
        prof_begin("qconv0", st);
        RUN(dense(w->qconv0, ws.f("l0"), 256, R, ws.f("q1"), 512, 1, st, rg(R_L0, N), nullptr, 0, 0, 3, N, 0, 0, EV2H_FAM_QCONV));
        prof_end("qconv0", st);
    }
    // ---- attention (TEHNet.py:13-27).  The second query convolution (Conv1d -> BN, affine) is folded behind the attention's sum
    // over the points (ev2h_attn_sim_folded): q2 is never formed.  (The unfolded form -- two more k = 3 GEMMs + ev2h_attn_sim -- is
    // what the oracle computes; the operators stay in the ABI and are tested against it, tests/test_gpu_ops.py.)
    if (fork) EV2H_CHECK_HIP(hipStreamWaitEvent((hipStream_t)st, side->ev[7], 0));              // logits ready
    if (zsum_fused) {
        RUN(ev2h_attn_simfold_partials(ws.f("zpart"), 128, ws.f("logits_pm"), B, N, w->qconv4T[0], w->qconv4T[1], w->qconv4[0].b, w->qconv4[1].b,
                                       ws.f("sim"), st));
    } else {
        RUN(ev2h_attn_sim_folded(ws.f("logits_pm"), ws.f("q1"), 512, B, N, w->qconv4T[0], w->qconv4T[1], w->qconv4[0].b, w->qconv4[1].b,
                                 ws.f("zpart"), ws.f("sim"), st));
    }
    if (l0_bf16) RUN(ev2h_attn_context_bf16rows(ws.f("sim"), ws.f("l0"), 256, B, N, ws.f("hf8"), w->l0_unscale, st));
    else if (l0_f16) RUN(ev2h_attn_context_f16rows(ws.f("sim"), ws.f("l0"), 256, B, N, ws.f("hf8"), ws.r(R_HF), B, w->l0_unscale, ws.p1scale(5), st));
    else RUN(ev2h_attn_context(ws.f("sim"), ws.f("l0"), 256, B, N, ws.f("hf8"), ws.r(R_HF), B, w->l0_unscale, st));
    // ---- MANO regressors (TEHNet.py:194-195, 68-112): left on the caller's stream, right on the side stream
    if (fork) {
        EV2H_CHECK_HIP(hipStreamWaitEvent((hipStream_t)st, side->ev[1], 0));       // ball queries done
        EV2H_CHECK_HIP(hipEventRecord(side->ev[2], (hipStream_t)st));              // attention output (hf8) ready
        EV2H_CHECK_HIP(hipStreamWaitEvent(side->stream, side->ev[2], 0));
    }
    for (int h = 0; h < 2; ++h) {
        const ev2h_sa_module& m = w->mano_sa1[h];
        const char* const* nm = kHandNames[h];
        ev2h_stream_t sh = (h == 1) ? sd : st;
        int32_t* gi[2] = {ws.i(nm[3]), ws.i(nm[4])};
        RUN(sa_module(prec_sa, h ? "manoR" : "manoL", m, ws.f("hf8") + (size_t)h * R * 8, 8, ws.f("pts4"), ws.f(nm[2]), gi, ws.i(nm[5]), B, N, ws.f(nm[0]), ws.f(nm[6]), 520, sh,
                      ws.r(R_HF + h), ws.r(R_P1M + h), ws.p1scale(2 + h), ws.r(R_M1 + h), 4, ws.f(nm[6]) + 512, 520));   // + the xyz columns of the regressor's sa2 input
        RUN(dense(w->mano_sa2[h][0], ws.f(nm[6]), 520, B * 128, ws.f(nm[7]), 256, 1, sh, rg(R_M1 + h, 128, R_MSA2H + h, 128, R_FEAT)));
        RUN(dense(w->mano_sa2[h][1], ws.f(nm[7]), 256, B * 128, ws.f(nm[8]), 512, 1, sh, rg(R_MSA2H + h, 128, R_M2 + h, 1), nullptr, 0, 0, 1, 0, 128));
        RUN(dense(w->head0[h], ws.f(nm[8]), 512, B, ws.f(nm[9]), 1024, 1, sh, rg(R_M2 + h, 1, R_FC1 + h, 1), nullptr, 0, 0, 1, 0, 0, 1));
        const int ldprm = out->params_stride ? (int)out->params_stride : w->head4[h].O;
        RUN(dense(w->head4[h], ws.f(nm[9]), 1024, B, out->params[h], ldprm, 0, sh, rg(R_FC1 + h, 1), nullptr, 0, 0, 1, 0, 0, 1));
        if (mano[h]) RUN(ev2h_mano(mano[h], out->params[h], ldprm, B, out->vertices[h], out->vertices_stride, out->joints[h], out->joints_stride, sh));
    }
    return EV2H_OK;
}

extern "C" int ev2h_forward(const ev2h_weights* w, const ev2h_mano_consts* mano_left, const ev2h_mano_consts* mano_right,
                            float* xyz_cm, int B, int C, int N, int mhlnes, const int64_t* fps_init, const ev2h_outputs* out,
                            void* workspace, size_t workspace_bytes, ev2h_stream_t st) {
    EV2H_CHECK_ARG(w && xyz_cm && fps_init && out && workspace);
    EV2H_CHECK_ARG(B > 0 && N >= 128 && N <= 32768 && (C == 4 || C == 5));    // size contract: see ev2hands_hip.h
    EV2H_CHECK_ARG(out->class_logits && out->params[0] && out->params[1]);
    // a NULL hand model skips that hand's MANO layer (the caller applies its own to params[h], TEHNet.py:103)
    EV2H_CHECK_ARG(!mano_left || (out->vertices[0] && out->joints[0]));
    EV2H_CHECK_ARG(!mano_right || (out->vertices[1] && out->joints[1]));
    EV2H_CHECK_ARG((out->logits_stride == 0 || out->logits_stride >= (size_t)4 * N) && (out->params_stride == 0 || out->params_stride >= (size_t)w->head4[0].O));
    // the head's width 3 + n_pose_params + 10 + 3 comes from the checkpoint (ev2h_pack_weights); a hand model must take that many PCA coefficients
    EV2H_CHECK_ARG(w->head4[0].O == w->head4[1].O && w->head4[0].O >= 17 && w->head4[0].O <= 61);
    for (int h = 0; h < 2; ++h) {
        const ev2h_mano_consts* mc = h ? mano_right : mano_left;
        if (mc && 3 + mc->ncomps + 13 != w->head4[h].O) {
            ev2h_set_error("ev2h_forward: the checkpoint regresses %d pose coefficients per hand, the %s MANO model takes %d", w->head4[h].O - 16,
                           h ? "right" : "left", mc->ncomps);
            return EV2H_ERR_ARG;
        }
    }
    EV2H_CHECK_ARG(out->params_stride <= 0x7fffffff);
    EV2H_CHECK_ARG((out->vertices_stride == 0 || out->vertices_stride >= 2334) && (out->joints_stride == 0 || out->joints_stride >= 63));
    EV2H_CHECK_ARG(w->sa1.npoint == 512 && w->sa2.npoint == 128 && w->mano_sa1[0].npoint == 128 && w->mano_sa1[1].npoint == 128);
    EV2H_CHECK_ARG(w->sa1.nbranch == 3 && w->sa2.nbranch == 2 && w->mano_sa1[0].nbranch == 2 && w->mano_sa1[1].nbranch == 2);
    if ((w->precision == EV2H_PREC_F16X2 || w->precision == EV2H_PREC_F16) && !(w->flags & (EV2H_W_EQUALIZED | EV2H_W_UNEQUALIZED_OK))) {
        // the F16X2 accuracy contract (ev2hands_hip.h): one power of two per window / per matrix keeps 22 bits only down to 2^-17 of
        // the maximum -- un-equalised checkpoints measured 6.6e-4 ... 0.23 relative error where equalised ones hold 1.3e-6
        ev2h_set_error("ev2h_forward: F16X2 / F16 weights are not channel-equalised (pack them with ev2h_pack_weights(..., EV2H_PACK_EQUALIZE), "
                       "or set EV2H_W_UNEQUALIZED_OK in ev2h_weights.flags to run them anyway, or use BF16X3 / F32)");
        return EV2H_ERR_ARG;
    }
    RUN(ev2h_init());
    g_precision = w->precision;
    g_f16_families = w->precision == EV2H_PREC_F16 ? w->f16_families : 0;
    Ws ws;
    ws.base = static_cast<char*>(workspace);
    ws.B = B;
    ws.ranges_on = (w->precision == EV2H_PREC_F16X2 || w->precision == EV2H_PREC_F16);      // the fp16-plane modes: range records are kept and used
    build_layout(ws.L, B, N);
    if (workspace_bytes < ws.L.total) {
        ev2h_set_error("ev2h_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, ws.L.total);
        return EV2H_ERR_WORKSPACE;
    }
    const ev2h_mano_consts* mano[2] = {mano_left, mano_right};
    g_side_claim = true;
    SideCtx* side = side_ctx(st);              // the side stream of THIS caller stream (forwards in flight on other streams have their own)
    g_side_claim = false;
    if (g_side_disabled) side = nullptr;
    bool forked = false;
    const int rc = forward_body(w, mano, xyz_cm, B, C, N, mhlnes, fps_init, out, ws, st, side, &forked);
    if (forked) {
        // join -- also on an error after the fork, so that the side stream is never left dangling (a stream capture of the
        // caller's stream would otherwise be invalidated by the un-joined fork)
        const hipError_t e1 = hipEventRecord(side->ev[3], side->stream);
        const hipError_t e2 = hipStreamWaitEvent((hipStream_t)st, side->ev[3], 0);
        if (rc == EV2H_OK && (e1 != hipSuccess || e2 != hipSuccess)) {
            ev2h_set_error("ev2h_forward: joining the side stream failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            return EV2H_ERR_HIP;
        }
    }
    return rc;
}
