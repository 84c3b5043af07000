// Per-frame joint metrics on the GPU (the consumer right after the hot path).
// Reference: /root/reference/src/Ev2Hands/evaluate.py:185-234 (absolute / relative / right-root-relative PCK curves),
// /root/reference/src/Ev2Hands/evaluate_ev2hands_r.py:35-89 (AUC = trapezoid / n rounded to 3 decimals, root-relative
// MPJPE, best-of-G ground-truth candidate by right-root-relative AUC, first one on ties).
// One wavefront per frame: lanes 0..41 own one joint each (hand = lane / 21); thresholds are tested with wave ballots.
// Distances are taken in float64 like the reference (float32 predictions * 1000 against float64 ground truth * 1000).
#include "common.hpp"
#include "ev2hands_hip.h"

namespace {

struct MetP {
    const float* left; const float* right;     // [B][21][3] metres
    const double* gts;                         // [B][G][2][21][3] metres
    int B, G, steps;
    double dist_max_mm;
    float* pck;                                // [B][3][steps+1]
    double* auc;                               // [B][3]  (unrounded)
    double* mpjpe; double* rootd;              // [B]
    int32_t* best;                             // [B]
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int lo = __shfl_xor(__double2loint(v), o, 64), hi = __shfl_xor(__double2hiint(v), o, 64);
        v += __hiloint2double(hi, lo);
    }
    return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int lo = __shfl_xor(__double2loint(v), o, 64), hi = __shfl_xor(__double2hiint(v), o, 64);
        v = fmin(v, __hiloint2double(hi, lo));
    }
    return v;
}

__global__ __launch_bounds__(64) void joint_metrics_kernel(MetP p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const bool act = lane < 42;
    const int hand = act ? lane / 21 : 0, j = act ? lane % 21 : 0;
    const float* src = (hand ? p.right : p.left) + ((size_t)b * 21 + j) * 3;
    // j3d_pred * 1000 and the root subtraction on the prediction are float32 tensor ops in the reference; the difference to
    // the float64 ground truth is then taken in float64
    const float fx = src[0] * 1000.f, fy = src[1] * 1000.f, fz = src[2] * 1000.f;
    const double px = (double)fx, py = (double)fy, pz = (double)fz;
    // roots: own hand (relative), right hand (right-root-relative)
    auto bcast = [&](double v, int from) {
        const int lo = __shfl(__double2loint(v), from, 64), hi = __shfl(__double2hiint(v), from, 64);
        return __hiloint2double(hi, lo);
    };
    const int own_root = hand * 21;
    // prediction relative to its own root / to the right root, formed in float32
    const double prelx = (double)(fx - __shfl(fx, own_root, 64)), prely = (double)(fy - __shfl(fy, own_root, 64)),
                 prelz = (double)(fz - __shfl(fz, own_root, 64));
    const double prrx = (double)(fx - __shfl(fx, 21, 64)), prry = (double)(fy - __shfl(fy, 21, 64)), prrz = (double)(fz - __shfl(fz, 21, 64));

    int best = 0;
    double best_auc = -1.0;
    const int n = p.steps + 1;
    for (int pass = 0; pass < 2; ++pass) {
        const int g0 = pass ? best : 0, g1 = pass ? best + 1 : p.G;
        for (int g = g0; g < g1; ++g) {
            const double* gs = p.gts + ((((size_t)b * p.G + g) * 2 + hand) * 21 + j) * 3;
            const double gx = gs[0] * 1000.0, gy = gs[1] * 1000.0, gz = gs[2] * 1000.0;
            const double grx = bcast(gx, own_root), gry = bcast(gy, own_root), grz = bcast(gz, own_root);
            const double gRx = bcast(gx, 21), gRy = bcast(gy, 21), gRz = bcast(gz, 21);
            double d[3];
            {   // absolute, root-relative, right-root-relative (differences formed exactly like the reference: (p - root) - (g - root))
                const double ax = px - gx, ay = py - gy, az = pz - gz;
                d[0] = sqrt(ax * ax + ay * ay + az * az);
                const double rx = prelx - (gx - grx), ry = prely - (gy - gry), rz = prelz - (gz - grz);
                d[1] = sqrt(rx * rx + ry * ry + rz * rz);
                const double qx = prrx - (gx - gRx), qy = prry - (gy - gRy), qz = prrz - (gz - gRz);
                d[2] = sqrt(qx * qx + qy * qy + qz * qz);
            }
            if (pass == 0) {
                // right-root-relative AUC of this candidate: trapezoid over the PCK curve / n, rounded to 3 decimals
                double sum = 0.0;
                float prev = 0.f;
                for (int s = 0; s < n; ++s) {
                    const double thr = (p.dist_max_mm / p.steps) * s;
                    const int k = __popcll(__ballot(act && d[2] < thr));
                    const float v = (float)k / 42.f;
                    if (s) sum += ((double)v + (double)prev) * 0.5;
                    prev = v;
                }
                const double a = rint(sum / n * 1000.0) / 1000.0;
                if (a > best_auc) { best_auc = a; best = g; }
            } else {
                for (int t = 0; t < 3; ++t) {
                    double sum = 0.0;
                    float prev = 0.f;
                    for (int s = 0; s < n; ++s) {
                        const double thr = (p.dist_max_mm / p.steps) * s;
                        const int k = __popcll(__ballot(act && d[t] < thr));
                        const float v = (float)k / 42.f;
                        if (lane == 0) p.pck[((size_t)b * 3 + t) * n + s] = v;
                        if (s) sum += ((double)v + (double)prev) * 0.5;
                        prev = v;
                    }
                    if (lane == 0) p.auc[(size_t)b * 3 + t] = sum / n;
                }
                const double m = wave_sum_f64(act ? d[1] : 0.0) / 42.0;
                // root distance: min over joints of |gt_left[j] - gt_right[j]| for the chosen candidate
                const double* gl = p.gts + ((((size_t)b * p.G + g) * 2 + 0) * 21 + (lane % 21)) * 3;
                const double* gr = gl + 21 * 3;
                const double ex = (gl[0] - gr[0]) * 1.0, ey = gl[1] - gr[1], ez = gl[2] - gr[2];
                const double dd = sqrt((ex * 1000.0) * (ex * 1000.0) + (ey * 1000.0) * (ey * 1000.0) + (ez * 1000.0) * (ez * 1000.0));
                const double rmin = wave_min_f64(lane < 21 ? dd : 1e300);
                if (lane == 0) { p.mpjpe[b] = m; p.rootd[b] = rmin; p.best[b] = g; }
            }
        }
    }
}

}  // namespace

extern "C" int ev2h_joint_metrics(const float* j3d_left, const float* j3d_right, const double* j3d_gts, int B, int G, int num_steps,
                                  double dist_max_mm, float* pck, double* auc, double* mpjpe, double* root_distance, int32_t* best,
                                  ev2h_stream_t stream) {
    EV2H_CHECK_ARG(j3d_left && j3d_right && j3d_gts && pck && auc && mpjpe && root_distance && best);
    EV2H_CHECK_ARG(B > 0 && G > 0 && num_steps > 0 && dist_max_mm > 0);
    MetP p{j3d_left, j3d_right, j3d_gts, B, G, num_steps, dist_max_mm, pck, auc, mpjpe, root_distance, best};
    joint_metrics_kernel<<<B, 64, 0, (hipStream_t)stream>>>(p);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}
