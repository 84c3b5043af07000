// MANO layer for gfx950: PCA pose -> Rodrigues -> shape/pose blend shapes -> kinematic chain ->
// linear blend skinning -> 21 joints.  One 832-thread workgroup per (window, hand), a vertex per thread; the 16 joint
// transforms live in LDS, the blend-shape matrix is stored
// transposed ([145][2336]) so vertex reads are coalesced.
//
// Reference: /root/reference/src/Ev2Hands/model/utils.py:25-31 (SmplxAdapter.__call__) which calls
// the third-party manopth ManoLayer.forward (use_pca=True, ncomps=6, flat_hand_mean=False,
// axis-angle root) -- not vendored in the reference; algorithm restated in oracle/mano_oracle.py.
// Rotation formula: /root/reference/src/Ev2Hands/losses.py:14-51.
#include <cstdlib>
#include "common.hpp"
#include "ev2hands_hip.h"

namespace {

constexpr int NV = 778, NJ = 16, NB = 10, NP = 135, NCOEF = NB + NP, LDB = 2336;

struct ManoP {
    ev2h_mano_consts c;
    const float* params; int ldp;
    float* verts; float* joints;
    size_t verts_stride, joints_stride;       // floats between consecutive windows
    int parts;                                // workgroups per window
};

// level-ordered chain: parents of joint k (MANO kintree), -1 for the root
__constant__ int c_parent[NJ] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
__constant__ int c_joint_reorder[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

// Steps 1-2 of the layer, shared by mano_kernel and the debug kernel that exposes the rotation matrices:
// 1. full pose = [global_orient, hands_mean + pca_coeffs @ comps];  2. Rodrigues (quaternion route, theta + 1e-8 inside the norm,
// /root/reference/src/Ev2Hands/losses.py:14-51).  Contains one __syncthreads(); the caller synchronises before reading s_R.
__device__ __forceinline__ void mano_pose_and_rotations(const ev2h_mano_consts& c, const float* prm, int tid, float* s_pose, float (*s_R)[9]) {
    const int nc = c.ncomps;
    if (tid < 3) s_pose[tid] = prm[tid];
    else if (tid < 48) {
        const int t = tid - 3;
        float acc = 0.f;
        for (int k = 0; k < nc; ++k) acc = __fmaf_rn(prm[3 + k], c.comps[k * 45 + t], acc);
        s_pose[tid] = __fadd_rn(c.hands_mean[t], acc);
    }
    __syncthreads();
    if (tid < NJ) {
        const float x = s_pose[3 * tid], y = s_pose[3 * tid + 1], z = s_pose[3 * tid + 2];
        const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
        const float ang = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)));
        const float nx = x / ang, ny = y / ang, nz = z / ang;
        const float ha = ang * 0.5f;
        const float cs = cosf(ha), sn = sinf(ha);
        float qw = cs, qx = sn * nx, qy = sn * ny, qz = sn * nz;
        const float qn = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(qw, qw), __fmul_rn(qx, qx)), __fmul_rn(qy, qy)), __fmul_rn(qz, qz)));
        qw /= qn; qx /= qn; qy /= qn; qz /= qn;
        const float w2 = qw * qw, x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
        const float wx = qw * qx, wy = qw * qy, wz = qw * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
        float* R = s_R[tid];
        R[0] = __fsub_rn(__fsub_rn(__fadd_rn(w2, x2), y2), z2);
        R[1] = __fsub_rn(__fmul_rn(2.f, xy), __fmul_rn(2.f, wz));
        R[2] = __fadd_rn(__fmul_rn(2.f, wy), __fmul_rn(2.f, xz));
        R[3] = __fadd_rn(__fmul_rn(2.f, wz), __fmul_rn(2.f, xy));
        R[4] = __fsub_rn(__fadd_rn(__fsub_rn(w2, x2), y2), z2);
        R[5] = __fsub_rn(__fmul_rn(2.f, yz), __fmul_rn(2.f, wx));
        R[6] = __fsub_rn(__fmul_rn(2.f, xz), __fmul_rn(2.f, wy));
        R[7] = __fadd_rn(__fmul_rn(2.f, wx), __fmul_rn(2.f, yz));
        R[8] = __fadd_rn(__fsub_rn(__fsub_rn(w2, x2), y2), z2);
    }
}

// debug / parity: the 16 rotation matrices the layer uses for each window (tests/test_gpu_ops.py)
__global__ __launch_bounds__(64) void mano_rotations_kernel(ev2h_mano_consts c, const float* params, int ldp, float* rot) {
    __shared__ float s_pose[48];
    __shared__ float s_R[NJ][9];
    const int b = blockIdx.x, tid = threadIdx.x;
    mano_pose_and_rotations(c, params + (size_t)b * ldp, tid, s_pose, s_R);
    __syncthreads();
    for (int i = tid; i < NJ * 9; i += 64) rot[(size_t)b * NJ * 9 + i] = s_R[i / 9][i % 9];
}

constexpr int MANO_THREADS = 832;     // 13 waves: one vertex per thread (778), so the 145-term blend sums of a window run side by side

__global__ __launch_bounds__(MANO_THREADS) void mano_kernel(ManoP p) {
    __shared__ float s_pose[48];
    __shared__ float s_R[NJ][9];
    __shared__ float s_coef[NCOEF + 3];
    __shared__ float s_J[NJ][3];
    __shared__ float s_G[NJ][12];      // rows of [R | t]
    __shared__ float s_A[NJ][12];
    __shared__ float s_tip[5][3];
    __shared__ float s_vp[NV * 3];     // posed-shape vertices of this workgroup's part
    // p.parts workgroups per (window, hand) at small batches: each repeats the (tiny) joint chain and skins a quarter of the vertices.
    // One workgroup pulls the 1.35 MB blend-shape matrix through ONE CU (46 us for a one-window forward, at its very end); a vertex's
    // arithmetic does not depend on who computes it.
    const int b = blockIdx.x / p.parts, part = blockIdx.x % p.parts, tid = threadIdx.x;
    const int vper = (NV + p.parts - 1) / p.parts, v_begin = part * vper, v_end = min(NV, v_begin + vper);
    const float* prm = p.params + (size_t)b * p.ldp;
    const int nc = p.c.ncomps;
    const float* betas = prm + 3 + nc;
    const float* transl = prm + 3 + nc + NB;

    mano_pose_and_rotations(p.c, prm, tid, s_pose, s_R);       // steps 1-2 (ends with the rotations written, not yet synchronised)
    if (tid >= 64 && tid < 64 + NB) s_coef[tid - 64] = betas[tid - 64];
    // 4. joints of the shaped template: J = J_template + J_shape^T beta
    if (tid >= 64 && tid < 64 + 48) {
        const int t = tid - 64;
        float acc = 0.f;
        for (int k = 0; k < NB; ++k) acc = __fmaf_rn(betas[k], p.c.J_shape[k * 48 + t], acc);
        s_J[t / 3][t % 3] = __fadd_rn(acc, p.c.J_template[t]);
    }
    __syncthreads();

    // 3. pose map (R_k - I for the 15 articulated joints) behind the 10 betas
    if (tid < NP) {
        const int e = tid % 9;
        s_coef[NB + tid] = __fsub_rn(s_R[1 + tid / 9][e], (e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    // 5. kinematic chain, root then three levels (joints 1,4,7,10,13 / 2,5,.. / 3,6,..)
    if (tid == 0) {
        for (int r = 0; r < 3; ++r) {
            s_G[0][4 * r + 0] = s_R[0][3 * r + 0]; s_G[0][4 * r + 1] = s_R[0][3 * r + 1]; s_G[0][4 * r + 2] = s_R[0][3 * r + 2];
            s_G[0][4 * r + 3] = s_J[0][r];
        }
    }
    __syncthreads();
    for (int lev = 1; lev <= 3; ++lev) {
        if (tid < 5) {
            const int k = 3 * tid + lev, par = c_parent[k];
            const float tx = __fsub_rn(s_J[k][0], s_J[par][0]), ty = __fsub_rn(s_J[k][1], s_J[par][1]),
                        tz = __fsub_rn(s_J[k][2], s_J[par][2]);
            const float* P = s_G[par];
            const float* R = s_R[k];
            for (int r = 0; r < 3; ++r) {
                const float p0 = P[4 * r], p1 = P[4 * r + 1], p2 = P[4 * r + 2], p3 = P[4 * r + 3];
                for (int c = 0; c < 3; ++c)
                    s_G[k][4 * r + c] = __fmaf_rn(p2, R[6 + c], __fmaf_rn(p1, R[3 + c], __fmul_rn(p0, R[c])));
                // 4x4 product, last column: p0*tx + p1*ty + p2*tz + p3*1
                s_G[k][4 * r + 3] = __fmaf_rn(p3, 1.f, __fmaf_rn(p2, tz, __fmaf_rn(p1, ty, __fmul_rn(p0, tx))));
            }
        }
        __syncthreads();
    }
    // 6. remove the rest-pose joint location: A = G with t - R_G j
    if (tid < NJ) {
        const float jx = s_J[tid][0], jy = s_J[tid][1], jz = s_J[tid][2];
        for (int r = 0; r < 3; ++r) {
            const float g0 = s_G[tid][4 * r], g1 = s_G[tid][4 * r + 1], g2 = s_G[tid][4 * r + 2];
            s_A[tid][4 * r] = g0; s_A[tid][4 * r + 1] = g1; s_A[tid][4 * r + 2] = g2;
            const float rj = __fmaf_rn(g2, jz, __fmaf_rn(g1, jy, __fmul_rn(g0, jx)));
            s_A[tid][4 * r + 3] = __fsub_rn(s_G[tid][4 * r + 3], rj);
        }
    }
    __syncthreads();

    // 7a. blend shapes: one thread per (vertex, coordinate) -- the 145-term sums of a vertex's three coordinates are independent, so
    //     they run side by side (the same sums in the same order as one thread per vertex computed them: bit-identical) and the
    //     column reads of consecutive threads are consecutive addresses
    for (int task = tid; task < (v_end - v_begin) * 3; task += MANO_THREADS) {
        const int vc = v_begin * 3 + task;                 // = v * 3 + c
        const float* col = p.c.blend_T + vc;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NB; ++k) s = __fmaf_rn(s_coef[k], col[(size_t)k * LDB], s);
        const float vs = __fadd_rn(s, p.c.v_template[vc]);
        float q = 0.f;
#pragma unroll 15
        for (int k = NB; k < NCOEF; ++k) q = __fmaf_rn(s_coef[k], col[(size_t)k * LDB], q);
        s_vp[task] = __fadd_rn(vs, q);
    }
    __syncthreads();
    // 7b. skinning per vertex
    const float trx = transl[0], try_ = transl[1], trz = transl[2];
    for (int v = v_begin + tid; v < v_end; v += MANO_THREADS) {
        const float vp[3] = {s_vp[(v - v_begin) * 3], s_vp[(v - v_begin) * 3 + 1], s_vp[(v - v_begin) * 3 + 2]};
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        const float* wv = p.c.weights + v * NJ;
        // (not unrolled further: fully unrolled, the compiler hoisted all 16 x 12 joint-transform values out of the vertex loop into
        //  registers -- 192 values against a 128-register budget, 396 bytes of scratch per lane -- for a loop that runs once)
#pragma unroll 2
        for (int k = 0; k < NJ; ++k) {
            const float w = wv[k];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = __fmaf_rn(s_A[k][e], w, T[e]);
        }
        float o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            o[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4 * r], vp[0]), __fmul_rn(T[4 * r + 1], vp[1])),
                                       __fmul_rn(T[4 * r + 2], vp[2])), T[4 * r + 3]);
#pragma unroll
        for (int t = 0; t < 5; ++t)
            if (v == p.c.tips[t]) { s_tip[t][0] = o[0]; s_tip[t][1] = o[1]; s_tip[t][2] = o[2]; }
        float* ov = p.verts + (size_t)b * p.verts_stride + v * 3;
        ov[0] = __fdiv_rn(__fmul_rn(__fadd_rn(o[0], trx), 1000.f), 1000.f);
        ov[1] = __fdiv_rn(__fmul_rn(__fadd_rn(o[1], try_), 1000.f), 1000.f);
        ov[2] = __fdiv_rn(__fmul_rn(__fadd_rn(o[2], trz), 1000.f), 1000.f);
    }
    __syncthreads();
    // 8. 16 chain joints + 5 fingertip vertices, reordered
    if (tid < 21) {
        const int src = c_joint_reorder[tid];
        // (chain joints: the first part; a fingertip: the part that skinned its vertex)
        // (p.c.tips is a kernel-argument array: a run-time index into it made the compiler copy the whole argument block to
        //  scratch -- 396 bytes per lane; the select chain keeps it in scalar registers)
        int tipv = -1;
#pragma unroll
        for (int t = 0; t < 5; ++t) tipv = (src - NJ == t) ? p.c.tips[t] : tipv;
        const bool mine = (src < NJ) ? part == 0 : (tipv >= v_begin && tipv < v_end);
        if (!mine) return;
        float j[3];
        for (int c = 0; c < 3; ++c) j[c] = (src < NJ) ? s_G[src][4 * c + 3] : s_tip[src - NJ][c];
        float* oj = p.joints + (size_t)b * p.joints_stride + tid * 3;
        oj[0] = __fdiv_rn(__fmul_rn(__fadd_rn(j[0], trx), 1000.f), 1000.f);
        oj[1] = __fdiv_rn(__fmul_rn(__fadd_rn(j[1], try_), 1000.f), 1000.f);
        oj[2] = __fdiv_rn(__fmul_rn(__fadd_rn(j[2], trz), 1000.f), 1000.f);
    }
}

}  // namespace

extern "C" int ev2h_mano(const ev2h_mano_consts* c, const float* params, int ldp, int B, float* verts, size_t verts_stride, float* joints,
                         size_t joints_stride, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(c && params && verts && joints && B > 0);
    EV2H_CHECK_ARG(c->hands_mean && c->comps && c->blend_T && c->v_template && c->J_template && c->J_shape && c->weights);
    EV2H_CHECK_ARG(c->ncomps >= 1 && c->ncomps <= 45 && ldp >= 3 + c->ncomps + 13);
    ManoP p{};
    EV2H_CHECK_ARG((verts_stride == 0 || verts_stride >= (size_t)NV * 3) && (joints_stride == 0 || joints_stride >= 63));
    p.c = *c; p.params = params; p.ldp = ldp; p.verts = verts; p.joints = joints;
    p.verts_stride = verts_stride ? verts_stride : (size_t)NV * 3; p.joints_stride = joints_stride ? joints_stride : 63;
    p.parts = B <= 32 ? 4 : 1;      // a few windows at a time: a hand's vertices over four workgroups (tests: B = 1 .. 64 against the oracle)
    mano_kernel<<<B * p.parts, MANO_THREADS, 0, (hipStream_t)stream>>>(p);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

extern "C" int ev2h_mano_rotations(const ev2h_mano_consts* c, const float* params, int ldp, int B, float* rot, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(c && params && rot && B > 0 && c->hands_mean && c->comps);
    EV2H_CHECK_ARG(c->ncomps >= 1 && c->ncomps <= 45 && ldp >= 3 + c->ncomps);
    mano_rotations_kernel<<<B, 64, 0, (hipStream_t)stream>>>(*c, params, ldp, rot);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}
