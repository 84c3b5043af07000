// Two-hand mesh self-collision on the GPU, the per-frame consumer of the predicted vertices:
//   /root/reference/src/Ev2Hands/evaluate_ev2hands_r.py:128-160  compute_non_collision_score
//   /root/reference/src/Ev2Hands/utils/__init__.py:106-124       compute_collision_percentage
// The reference hands `vertices[faces]` of the concatenated left+right mesh (float32 metres * 1000, widened to float64) to
// the BVH of the un-vendored torch-mesh-isect package and counts the returned triangle pairs.  That package's arithmetic
// is not available (parity unpinned, see oracle/collision_oracle.py); this kernel computes the quantity the tree
// approximates: ALL unordered pairs of triangles that share no vertex index and intersect (separating-axis test in
// float64), in lexicographic order, no per-triangle cap.
//
// 3 076 triangles per window are 4.7 M pairs: no tree is needed.  One 1024-thread workgroup per window keeps the window's
// vertices (float32, exact), faces and float32 bounding boxes in LDS (142 KB); each wave owns 64-row blocks of the pair
// matrix and walks the columns j uniformly, so box j and face j are LDS broadcasts; only box-overlapping, non-adjacent
// candidates reach the float64 test.  Candidates are queued per wave (LDS, in column order) and tested 64 at a time, one
// per lane -- with one candidate per lane-ROW the long float64 test would run for every column that has any candidate, a few
// lanes at a time (meshes that intersect themselves everywhere made that 12.7 ms per 256 windows; queued: 2.4 ms)
// -- and their verdicts are applied in queue order by the lane that owns the row, so the accepted set, the per-triangle cap and
// the pair order are exactly those of the sequential walk.  Two passes (count, prefix sum over the rows, write) make the pair
// list deterministic.
#include <cstdlib>
#include "common.hpp"
#include "ev2hands_hip.h"

namespace {

constexpr int COL_MAX_V = 778, COL_MAX_F = 1538, COL_THREADS = 1024;

struct ColP {
    const float* vl; const float* vr;          // [B][nv][3] metres
    const int32_t* fl; const int32_t* fr;      // [nf][3]
    int nv, nf;
    float scale;
    int max_pairs;
    int32_t* pairs;                            // [B][max_pairs][2] or null
    int32_t* counts;                           // [B]
    int cap;                                   // per-triangle cap on recorded pairs (the BVH's max_collisions), 0 = none
    // Several workgroups per window (ev2h_mesh_collisions_ws): workgroup w of nsplit takes every nsplit-th wave's row blocks;
    // the per-row counts then meet in global memory (rowcnt [B][2 nf + 1]) and a second launch (phase 1) forms the prefix sums,
    // the window's count and the compacted list; without the one-walk list a third launch (phase 2) writes the pairs.
    int nsplit, phase;
    int32_t* rowcnt;
};

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// true when `ax` separates the triangles (projections strictly disjoint); degenerate axes never separate
__device__ __forceinline__ bool separated(const V3 (&a)[3], const V3 (&b)[3], V3 ax) {
    if (dot(ax, ax) < 1e-20) return false;
    const double a0 = dot(a[0], ax), a1 = dot(a[1], ax), a2 = dot(a[2], ax);
    const double b0 = dot(b[0], ax), b1 = dot(b[1], ax), b2 = dot(b[2], ax);
    const double amin = fmin(a0, fmin(a1, a2)), amax = fmax(a0, fmax(a1, a2));
    const double bmin = fmin(b0, fmin(b1, b2)), bmax = fmax(b0, fmax(b1, b2));
    return amax < bmin || bmax < amin;
}

__device__ bool sat_intersect(const V3 (&a)[3], const V3 (&b)[3]) {
    const V3 ea[3] = {sub(a[1], a[0]), sub(a[2], a[1]), sub(a[0], a[2])};
    const V3 eb[3] = {sub(b[1], b[0]), sub(b[2], b[1]), sub(b[0], b[2])};
    const V3 na = cross(ea[0], ea[1]), nb = cross(eb[0], eb[1]);
    if (separated(a, b, na) || separated(a, b, nb)) return false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (separated(a, b, cross(ea[i], eb[j]))) return false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (separated(a, b, cross(na, ea[i])) || separated(a, b, cross(nb, eb[i]))) return false;
    return true;
}

// SPLIT / PHASE: the call forms are separate instantiations (one workgroup per window: <false, 0>; two per window: <true, 0> counting
// walk, <true, 1> prefix sums + compaction, <true, 2> list walk) -- one body with run-time tests of p.nsplit / p.phase sat at the
// 128-register cap of a 1024-thread workgroup with 19 registers in scratch
template <bool SPLIT, int PHASE>
__global__ __launch_bounds__(COL_THREADS) void mesh_collision_kernel(ColP p) {
    static_assert(SPLIT || PHASE == 0, "phases belong to the split form");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int V2 = 2 * p.nv, F2 = 2 * p.nf;
    float* sv = reinterpret_cast<float*>(smem);                 // [V2][3] mm
    int* sf = reinterpret_cast<int*>(sv + 3 * V2);              // [F2][3]
    float* sbb = reinterpret_cast<float*>(sf + 3 * F2);         // [F2][6] min xyz, max xyz
    int* srow = reinterpret_cast<int*>(sbb + 6 * F2);           // [F2 + 1] pairs per row, then exclusive prefix
    int* spart = srow + F2 + 1;                                 // [COL_THREADS] scan scratch
    int* squeue = spart + COL_THREADS + (threadIdx.x >> 6) * 128;   // [waves][128] this wave's candidates: (lane << 16) | column
    float* sblk = reinterpret_cast<float*>(spart + COL_THREADS + (COL_THREADS >> 6) * 128);   // [nblk][6] box of every 64-triangle block
    const int nsplit = SPLIT ? p.nsplit : 1;
    const int wgs = PHASE == 1 ? 1 : nsplit;                  // (phase 1 runs one workgroup per window)
    const int b = blockIdx.x / wgs, wg = blockIdx.x % wgs, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* grow = p.rowcnt ? p.rowcnt + (size_t)b * (F2 + 1) : nullptr;

    if constexpr (PHASE != 1)
    for (int i = tid; i < 3 * V2; i += COL_THREADS) {
        const int v = i / 3, c = i - 3 * v;
        const float x = v < p.nv ? p.vl[((size_t)b * p.nv + v) * 3 + c] : p.vr[((size_t)b * p.nv + (v - p.nv)) * 3 + c];
        sv[i] = __fmul_rn(x, p.scale);                          // float32 multiply like `.numpy() * 1000`
    }
    if constexpr (PHASE != 1)
    for (int i = tid; i < 3 * F2; i += COL_THREADS) {
        const int f = i / 3, c = i - 3 * f;
        sf[i] = f < p.nf ? p.fl[f * 3 + c] : p.fr[(f - p.nf) * 3 + c] + p.nv;
    }
    __syncthreads();
    if constexpr (PHASE != 1)
    for (int f = tid; f < F2; f += COL_THREADS) {
        float lo[3], hi[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x0 = sv[3 * sf[3 * f] + c], x1 = sv[3 * sf[3 * f + 1] + c], x2 = sv[3 * sf[3 * f + 2] + c];
            lo[c] = fminf(x0, fminf(x1, x2));
            hi[c] = fmaxf(x0, fmaxf(x1, x2));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { sbb[6 * f + c] = lo[c]; sbb[6 * f + 3 + c] = hi[c]; }
    }
    __syncthreads();

    // Two-level culling: the bounding box of every block of 64 consecutive triangles.  A (row block, column block) pair whose
    // boxes are disjoint cannot hold an overlapping triangle pair and is skipped as a whole -- conservative, so the candidates,
    // their order and therefore the counts, caps and pair lists are exactly those of the full walk.  Mesh faces are stored
    // locally coherent (a block is a patch of the surface), and the two hands are usually apart: a hand-like pair of meshes
    // keeps ~1/8 of the block pairs (profiles/r3_collision_timing.txt); random triangle soup keeps all of them.
    if constexpr (PHASE != 1) {
        const int nblk_ = (F2 + 63) >> 6;
        for (int kb = wave; kb < nblk_; kb += (COL_THREADS >> 6)) {
            const int f = kb * 64 + lane;
            float v[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) { v[c] = f < F2 ? sbb[6 * f + c] : INFINITY; v[3 + c] = f < F2 ? sbb[6 * f + 3 + c] : -INFINITY; }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
                for (int c = 0; c < 3; ++c) { v[c] = fminf(v[c], __shfl_xor(v[c], o, 64)); v[3 + c] = fmaxf(v[3 + c], __shfl_xor(v[3 + c], o, 64)); }
            if (lane < 6) sblk[6 * kb + lane] = v[lane];
        }
    }
    __syncthreads();

    auto tri = [&](const int (&f)[3], V3 (&t)[3]) {
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = {(double)sv[3 * f[k]], (double)sv[3 * f[k] + 1], (double)sv[3 * f[k] + 2]};
    };
    const int nblk = (F2 + 63) >> 6, nwaves = COL_THREADS >> 6;
    int32_t* out = p.pairs ? p.pairs + (size_t)b * p.max_pairs * 2 : nullptr;
    // ONE walk when every row's pairs fit a fixed slot range of the output (a per-triangle cap and a list of >= F2 * cap entries, the
    // default capacity of ev2hands_amd.collision.CollisionLoss): row i writes its accepted pairs at slots [i * cap, i * cap + cnt_i)
    // during the counting walk, and the list is compacted in place afterwards (destinations never lie behind their sources, rows
    // ascending: a chunk is read into registers, then written).  Same acceptance order, hence the same list as the two-walk form,
    // for half the separating-axis tests.
    const bool one_walk = out && p.cap > 0 && (long)p.max_pairs >= (long)F2 * p.cap;

    // Row blocks are dealt to the waves (of all the window's workgroups) in snake order: block rb scans the column blocks rb .. nblk-1,
    // so dealing them round-robin gave the first wave 100 column blocks and the last 54 (49 row blocks, 16 waves); boustrophedon
    // rounds level that (84 / 70), and with two workgroups per window no wave has more than block 0's 49.  A row's result does not
    // depend on who computes it.
    const int NW = nwaves * nsplit, gw = wg * nwaves + wave;
    auto walk = [&](int pass) {
        for (int rnd = 0; rnd * NW < nblk; ++rnd) {
            const int rb = rnd * NW + ((rnd & 1) ? NW - 1 - gw : gw);
            if (rb >= nblk) continue;
            const int i = rb * 64 + lane;
            const bool valid = i < F2;
            const int ii = valid ? i : F2 - 1;
            float lo[3], hi[3];
            int fi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { lo[c] = sbb[6 * ii + c]; hi[c] = sbb[6 * ii + 3 + c]; fi[c] = sf[3 * ii + c]; }
            int cnt = 0;
            const int base = one_walk ? ii * p.cap : (pass ? srow[ii] : 0);
            int qn = 0;                                         // queued candidates (wave-uniform)
            // the first n queued candidates: one float64 test per lane, then every row's owner applies its verdicts in queue order
            auto drain = [&](int n) {
                bool hit = false;
                if (lane < n) {
                    const int e = squeue[lane];
                    const int ci = rb * 64 + (e >> 16), cj = e & 0xffff;
                    const int fa[3] = {sf[3 * ci], sf[3 * ci + 1], sf[3 * ci + 2]}, fb[3] = {sf[3 * cj], sf[3 * cj + 1], sf[3 * cj + 2]};
                    V3 ta[3], tb[3];
                    tri(fa, ta);
                    tri(fb, tb);
                    hit = sat_intersect(ta, tb);
                }
                const unsigned long long hm = __ballot(hit);
                for (int k = 0; k < n; ++k) {
                    const int e = squeue[k];                    // LDS broadcast
                    if (((hm >> k) & 1ull) && (e >> 16) == lane && (p.cap <= 0 || cnt < p.cap)) {   // at most `cap` pairs per triangle i, in j order
                        if ((pass || one_walk) && out && base + cnt < p.max_pairs) { out[2 * (base + cnt)] = i; out[2 * (base + cnt) + 1] = e & 0xffff; }
                        ++cnt;
                    }
                }
                const int rest = qn - n;                        // < 64: move the tail to the front
                const int keep = (lane < rest) ? squeue[n + lane] : 0;
                if (lane < rest) squeue[lane] = keep;
                qn = rest;
            };
            const float* rbx = sblk + 6 * rb;
            for (int jb = rb; jb < nblk; ++jb) {                // column blocks in order; disjoint block boxes are skipped whole
                const float* cbx = sblk + 6 * jb;
                if (!(rbx[0] <= cbx[3] && cbx[0] <= rbx[3] && rbx[1] <= cbx[4] && cbx[1] <= rbx[4] && rbx[2] <= cbx[5] && cbx[2] <= rbx[5])) continue;
            const int j_end = min(F2, jb * 64 + 64);
            for (int j = max(jb * 64, rb * 64 + 1); j < j_end; ++j) {            // wave-uniform column: LDS broadcasts
                const float* bj = sbb + 6 * j;
                bool ov = valid && j > i && (p.cap <= 0 || cnt < p.cap) && lo[0] <= bj[3] && bj[0] <= hi[0] && lo[1] <= bj[4] &&
                          bj[1] <= hi[1] && lo[2] <= bj[5] && bj[2] <= hi[2];
                if (ov) {
                    const int fj[3] = {sf[3 * j], sf[3 * j + 1], sf[3 * j + 2]};
#pragma unroll
                    for (int x = 0; x < 3; ++x)
#pragma unroll
                        for (int y = 0; y < 3; ++y) ov = ov && (fi[x] != fj[y]);         // triangles that share a vertex are not tested
                }
                const unsigned long long m = __ballot(ov);
                if (m) {
                    const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (ov) squeue[pos] = (lane << 16) | j;
                    qn += __popcll(m);
                    if (qn >= 64) drain(64);
                }
            }
            }
            while (qn > 0) drain(min(qn, 64));
            if (!pass && valid) { if constexpr (SPLIT) grow[i] = cnt; else srow[i] = cnt; }
        }
    };
    if constexpr (SPLIT && PHASE == 0) { walk(0); return; }
    if constexpr (SPLIT && PHASE == 2) {                         // bases = the exclusive prefix phase 1 left in rowcnt
        for (int r = tid; r <= F2; r += COL_THREADS) srow[r] = grow[r];
        __syncthreads();
        walk(1);
        return;
    }
    if constexpr (SPLIT) {                                       // phase 1: the row counts of all workgroups
        for (int r = tid; r < F2; r += COL_THREADS) srow[r] = grow[r];
    }
    for (int pass = 0; pass < 2; ++pass) {
        if constexpr (!SPLIT) walk(pass);
        if (pass) break;
        __syncthreads();
        // exclusive prefix sum of srow[0..F2): each thread sums a contiguous chunk, Hillis-Steele over the chunk totals
        const int per = (F2 + COL_THREADS - 1) / COL_THREADS;
        const int lo_r = min(tid * per, F2), hi_r = min(lo_r + per, F2);
        int s = 0;
        for (int r = lo_r; r < hi_r; ++r) s += srow[r];
        spart[tid] = s;
        __syncthreads();
        for (int o = 1; o < COL_THREADS; o <<= 1) {
            const int add = tid >= o ? spart[tid - o] : 0;
            __syncthreads();
            spart[tid] += add;
            __syncthreads();
        }
        int run = spart[tid] - s;
        for (int r = lo_r; r < hi_r; ++r) { const int c = srow[r]; srow[r] = run; run += c; }
        if (tid == COL_THREADS - 1) { p.counts[b] = spart[tid]; srow[F2] = spart[tid]; }
        if (!out) break;                                        // counts only: the second walk would repeat every test for nothing
        __syncthreads();
        if (SPLIT && !one_walk) {                                // phase 2 (another launch) writes the pairs at these bases
            for (int r = tid; r <= F2; r += COL_THREADS) grow[r] = srow[r];
            break;
        }
        if (one_walk) {
            // in-place compaction of the per-row slot ranges: entry e = (row r = e / cap, k = e % cap) is live iff k < cnt_r and moves
            // to srow[r] + k <= e
            const int total_slots = F2 * p.cap;
            for (int e0 = 0; e0 < total_slots; e0 += COL_THREADS) {
                const int e = e0 + tid;
                int a = 0, c = 0, dst = -1;
                if (e < total_slots) {
                    const int r = e / p.cap, k = e - r * p.cap;
                    if (k < srow[r + 1] - srow[r]) { dst = srow[r] + k; a = out[2 * e]; c = out[2 * e + 1]; }
                }
                __syncthreads();
                if (dst >= 0 && dst != e) { out[2 * dst] = a; out[2 * dst + 1] = c; }
                __syncthreads();
            }
            break;
        }
    }
}

// ---------------------------------------------------------------------------------------- penetration penalty
// Conic distance-field penalty of the colliding pairs (losses.py:60-102: CollisionLoss -> torch-mesh-isect's
// DistanceFieldPenetrationLoss(sigma = 0.5, point2plane = False, penalize_outside = False); un-vendored, restated from the
// published definition, Tzionas et al., IJCV 2016, eq. 11-14 -- parity unpinned, oracle/collision_oracle.py).  For a triangle f
// with circumcentre o, circumradius r and unit normal n, a point v at depth h = -n.(v - o) >= 0 behind the face lies in a cone
// whose radius grows as r (1 + h / sigma);  Phi = |(v - o) + h n| / (r (1 + h / sigma));  Psi = (1 - Phi)^2 if Phi < 1 (and the
// point is not in front of the face), else 0;  a pair (i, j) costs  sum_{v in j} Psi_i(v)^2 + sum_{v in i} Psi_j(v)^2.
struct PenP {
    const float* vl; const float* vr; const int32_t* fl; const int32_t* fr;
    int nv, nf; float scale; double sigma;
    const int32_t* pairs; const int32_t* counts; int max_pairs;
    double* loss;
};

__device__ double cone_term(const V3 (&f)[3], const V3 (&q)[3], double sigma) {
    const V3 a = sub(f[1], f[0]), b = sub(f[2], f[0]);
    const V3 axb = cross(a, b);
    const double n2 = dot(axb, axb);
    if (n2 < 1e-300) return 0.0;                                 // degenerate face: no cone
    // circumcentre o = f0 + ((|a|^2 b - |b|^2 a) x (a x b)) / (2 |a x b|^2);  circumradius r = |a| |b| |a - b| / (2 |a x b|)
    const double a2 = dot(a, a), b2 = dot(b, b);
    const V3 t = {a2 * b.x - b2 * a.x, a2 * b.y - b2 * a.y, a2 * b.z - b2 * a.z};
    const V3 c = cross(t, axb);
    const V3 o = {f[0].x + c.x / (2 * n2), f[0].y + c.y / (2 * n2), f[0].z + c.z / (2 * n2)};
    const V3 amb = sub(a, b);
    const double r = sqrt(a2) * sqrt(b2) * sqrt(dot(amb, amb)) / (2 * sqrt(n2));
    const double inv = 1.0 / sqrt(n2);
    const V3 n = {axb.x * inv, axb.y * inv, axb.z * inv};
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const V3 d = sub(q[k], o);
        const double along = dot(d, n);                         // > 0: in front of the face (penalize_outside = False: free)
        if (along > 0.0) continue;
        const V3 rad = {d.x - along * n.x, d.y - along * n.y, d.z - along * n.z};
        const double phi = sqrt(dot(rad, rad)) / (r * (1.0 - along / sigma));
        if (phi < 1.0) { const double psi = (1.0 - phi) * (1.0 - phi); s += psi * psi; }
    }
    return s;
}

__global__ __launch_bounds__(256) void collision_penalty_kernel(PenP p) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(p.counts[b], p.max_pairs);
    const int32_t* pr = p.pairs + (size_t)b * p.max_pairs * 2;
    auto vert = [&](int v) -> V3 {
        const float* src = v < p.nv ? p.vl + ((size_t)b * p.nv + v) * 3 : p.vr + ((size_t)b * p.nv + (v - p.nv)) * 3;
        return {(double)__fmul_rn(src[0], p.scale), (double)__fmul_rn(src[1], p.scale), (double)__fmul_rn(src[2], p.scale)};
    };
    auto tri = [&](int f, V3 (&t)[3]) {
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = vert(f < p.nf ? p.fl[f * 3 + k] : p.fr[(f - p.nf) * 3 + k] + p.nv);
    };
    double s = 0.0;
    for (int q = tid; q < n; q += 256) {
        V3 ti[3], tj[3];
        tri(pr[2 * q], ti);
        tri(pr[2 * q + 1], tj);
        s += cone_term(ti, tj, p.sigma) + cone_term(tj, ti, p.sigma);
    }
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {                          // fixed tree: deterministic
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) p.loss[b] = red[0];
}

}  // namespace

extern "C" int ev2h_collision_penalty(const float* verts_left, const float* verts_right, const int32_t* faces_left,
                                      const int32_t* faces_right, int B, int nv, int nf, float scale, double sigma, const int32_t* pairs,
                                      const int32_t* counts, int max_pairs, double* loss, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(verts_left && verts_right && faces_left && faces_right && pairs && counts && loss);
    EV2H_CHECK_ARG(B > 0 && nv >= 3 && nf >= 1 && max_pairs > 0 && sigma > 0.0);
    PenP p{verts_left, verts_right, faces_left, faces_right, nv, nf, scale, sigma, pairs, counts, max_pairs, loss};
    collision_penalty_kernel<<<B, 256, 0, (hipStream_t)stream>>>(p);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" size_t ev2h_mesh_collisions_scratch_bytes(int B, int nf) {
    return (B > 0 && nf > 0) ? (size_t)B * (2 * (size_t)nf + 1) * sizeof(int32_t) : 0;
}

extern "C" int ev2h_mesh_collisions_ws(const float* verts_left, const float* verts_right, const int32_t* faces_left,
                                       const int32_t* faces_right, int B, int nv, int nf, float scale, int max_pairs, int32_t* pairs,
                                       int32_t* counts, int max_per_triangle, void* scratch, size_t scratch_bytes, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(verts_left && verts_right && faces_left && faces_right && counts);
    EV2H_CHECK_ARG(B > 0 && nv >= 3 && nv <= COL_MAX_V && nf >= 1 && nf <= COL_MAX_F && max_pairs >= 0 && (pairs || max_pairs == 0));
    EV2H_CHECK_ARG(max_per_triangle >= 0);
    ColP p{verts_left, verts_right, faces_left, faces_right, nv, nf, scale, max_pairs, pairs, counts, max_per_triangle, 1, 0, nullptr};
    const size_t lds = (size_t)(3 * 2 * nv) * 4 + (size_t)(3 * 2 * nf) * 4 + (size_t)(6 * 2 * nf) * 4 + (size_t)(2 * nf + 1) * 4 +
                       COL_THREADS * 4 + (COL_THREADS / 64) * 128 * 4 + (size_t)((2 * nf + 63) / 64) * 6 * 4;
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_collision_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_collision_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_collision_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_collision_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)););
    // One 1024-thread workgroup (142 KB of LDS) per window fills one CU: below 256 windows CUs idle (BASELINE config 5 runs B = 128 per
    // GPU).  With a scratch buffer a window's row blocks are split over two workgroups (more cannot help: row block 0 alone scans all
    // 49 column blocks).  The caller chooses: no scratch buffer = one workgroup per window.
    const int nsplit = (scratch && scratch_bytes >= ev2h_mesh_collisions_scratch_bytes(B, nf) && B <= 128) ? 2 : 1;
    if (nsplit == 1) {
        mesh_collision_kernel<false, 0><<<B, COL_THREADS, lds, (hipStream_t)stream>>>(p);
        EV2H_CHECK_LAUNCH();
        return EV2H_OK;
    }
    p.nsplit = nsplit;
    p.rowcnt = static_cast<int32_t*>(scratch);
    p.phase = 0;
    mesh_collision_kernel<true, 0><<<B * nsplit, COL_THREADS, lds, (hipStream_t)stream>>>(p);
    EV2H_CHECK_LAUNCH();
    p.phase = 1;
    mesh_collision_kernel<true, 1><<<B, COL_THREADS, lds, (hipStream_t)stream>>>(p);
    EV2H_CHECK_LAUNCH();
    const bool one_walk = pairs && max_per_triangle > 0 && (long)max_pairs >= (long)(2 * nf) * max_per_triangle;
    if (pairs && !one_walk) {
        p.phase = 2;
        mesh_collision_kernel<true, 2><<<B * nsplit, COL_THREADS, lds, (hipStream_t)stream>>>(p);
        EV2H_CHECK_LAUNCH();
    }
    return EV2H_OK;
}

extern "C" int ev2h_mesh_collisions(const float* verts_left, const float* verts_right, const int32_t* faces_left,
                                    const int32_t* faces_right, int B, int nv, int nf, float scale, int max_pairs, int32_t* pairs,
                                    int32_t* counts, int max_per_triangle, ev2h_stream_t stream) {
    return ev2h_mesh_collisions_ws(verts_left, verts_right, faces_left, faces_right, B, nv, nf, scale, max_pairs, pairs, counts, max_per_triangle,
                                   nullptr, 0, stream);
}
