// ev2h_pack_weights: checkpoint -> the packed device weights ev2h_forward takes (host code; include/ev2hands_hip.h "weight packing").
//
// What the reference does with nn.Module.load_state_dict + eval-mode BatchNorm layers at every forward
// (/root/reference/src/Ev2Hands/model/model.py:14-23, demo.py:83-84; BN placement: model/pointnet2_utils.py:198,256,314,
// model/TEHNet.py:49-55,135-141,150-166) is done here ONCE: BatchNorm folded in float64, hidden channels equalised by exact
// powers of two (the F16X2 accuracy contract), weights laid out, split into operand planes and uploaded.  All arithmetic is
// plain IEEE double in a fixed order (the library is built with -ffp-contract=off), so the result is reproducible byte for byte
// (tests/test_pack_abi.py compares every array with an independent numpy restatement, tests/ref_pack.py).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "common.hpp"
#include "ev2hands_hip.h"

namespace {

typedef std::vector<double> Vec;

[[noreturn]] void fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw std::runtime_error(buf);
}

inline int up(int x, int m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------------------- network schema
// (TEHNet.py:116-166, pointnet2_utils.py:161-275; the widths are the reference's constructor arguments)
struct MsgSpec { int nb; int mlp[3][3]; double radius[3]; int nsample[3]; int npoint; };
const MsgSpec SA1 = {3, {{32, 32, 64}, {64, 64, 128}, {64, 96, 128}}, {0.1, 0.2, 0.4}, {32, 64, 128}, 512};
const MsgSpec SA2 = {2, {{128, 128, 256}, {128, 196, 256}, {0, 0, 0}}, {0.4, 0.8, 0.0}, {64, 128, 0}, 128};
const MsgSpec MANO_SA1 = SA2;
const int SA3_MLP[3] = {256, 512, 1024}, FP3_MLP[2] = {256, 256}, FP2_MLP[2] = {256, 128}, FP1_MLP[3] = {128, 128, 256};
const int MANO_SA2_MLP[2] = {256, 512};
const int N_CLASSES = 4;
// the regression head's width 3 + n_pose_params + 10 + 3 (TEHNet.py:49-55,87-90) is read off the checkpoint: TEHNet(n_pose_params)
// takes any number of MANO PCA coefficients (TEHNet.py:114-125; the MANO layer has at most 45)
const int MANO_OUT_MIN = 3 + 1 + 10 + 3, MANO_OUT_MAX = 3 + 45 + 10 + 3;
const char* const SIDES[2] = {"left", "right"};
const double BN_EPS = 1e-5;

// ---------------------------------------------------------------------------------------------------- checkpoint access
struct Tensor { Vec v; std::vector<int64_t> shape; bool used = false; };

struct Ckpt {
    std::map<std::string, Tensor> t;
    // first dimension of a tensor (the widths that the checkpoint, not the schema, decides), 0 if the key is absent (get() reports it)
    int64_t dim0(const std::string& key) const {
        auto it = t.find(key);
        return (it == t.end() || it->second.shape.empty()) ? 0 : it->second.shape[0];
    }
    const Vec& get(const std::string& key, std::initializer_list<int64_t> shape) {
        auto it = t.find(key);
        if (it == t.end()) fail("ev2h_pack_weights: missing key \"%s\" in the checkpoint", key.c_str());
        Tensor& x = it->second;
        std::vector<int64_t> want(shape);
        if (x.shape != want) {
            std::string a, b;
            for (auto d : want) a += std::to_string(d) + ",";
            for (auto d : x.shape) b += std::to_string(d) + ",";
            fail("ev2h_pack_weights: size mismatch for \"%s\": expected [%s] got [%s]", key.c_str(), a.c_str(), b.c_str());
        }
        x.used = true;
        return x.v;
    }
};

// one folded layer: W [O][I][T] (T = taps, 1 or 3) in the checkpoint's column order, b [O], optional post-ReLU affine
struct Layer {
    int O = 0, I = 0, T = 1;
    Vec W, b, ps, pt;
    bool post = false;
};
typedef std::map<std::string, Layer> Folded;

void bn_affine(Ckpt& ck, const std::string& p, int C, Vec& alpha, Vec& beta) {
    const Vec &w = ck.get(p + ".weight", {C}), &bias = ck.get(p + ".bias", {C}), &mean = ck.get(p + ".running_mean", {C}),
              &var = ck.get(p + ".running_var", {C});
    alpha.resize(C); beta.resize(C);
    for (int c = 0; c < C; ++c) {
        alpha[c] = w[c] / std::sqrt(var[c] + BN_EPS);
        const double ma = mean[c] * alpha[c];
        beta[c] = bias[c] - ma;
    }
}

// conv weight of shape [O, I, tail...] (tail: {1,1}, {1}, {3} or none)
const Vec& conv_w(Ckpt& ck, const std::string& key, int O, int I, int tail_dims, int T) {
    if (tail_dims == 2) return ck.get(key, {O, I, 1, 1});
    if (tail_dims == 1) return ck.get(key, {O, I, T});
    return ck.get(key, {O, I});
}

Layer fold_conv_bn(Ckpt& ck, const std::string& pc, const std::string& pb, int O, int I, int tail_dims, int T = 1) {      // Conv -> BN
    Layer L; L.O = O; L.I = I; L.T = T;
    const Vec& W = conv_w(ck, pc + ".weight", O, I, tail_dims, T);
    const Vec& b = ck.get(pc + ".bias", {O});
    Vec alpha, beta;
    bn_affine(ck, pb, O, alpha, beta);
    L.W.resize(W.size()); L.b.resize(O);
    const size_t row = (size_t)I * T;
    for (int o = 0; o < O; ++o) {
        for (size_t k = 0; k < row; ++k) L.W[o * row + k] = W[o * row + k] * alpha[o];
        const double ab = alpha[o] * b[o];
        L.b[o] = ab + beta[o];
    }
    return L;
}

Layer post_conv(Ckpt& ck, const std::string& pc, const std::string& pb, int O, int I, int tail_dims, int T = 1) {          // Conv -> ReLU -> BN
    Layer L; L.O = O; L.I = I; L.T = T; L.post = true;
    L.W = conv_w(ck, pc + ".weight", O, I, tail_dims, T);
    L.b = ck.get(pc + ".bias", {O});
    bn_affine(ck, pb, O, L.ps, L.pt);
    return L;
}

Layer plain_conv(Ckpt& ck, const std::string& pc, int O, int I, int tail_dims) {
    Layer L; L.O = O; L.I = I; L.T = 1;
    L.W = conv_w(ck, pc + ".weight", O, I, tail_dims, 1);
    L.b = ck.get(pc + ".bias", {O});
    return L;
}

std::string idx(const std::string& p, int i) { return p + "." + std::to_string(i); }
std::string idx(const std::string& p, int i, int j) { return p + "." + std::to_string(i) + "." + std::to_string(j); }

void fold_msg(Ckpt& ck, Folded& F, const std::string& prefix, const MsgSpec& sp, int fan_in) {
    for (int i = 0; i < sp.nb; ++i) {
        int last = fan_in;
        for (int j = 0; j < 3; ++j) {
            F[idx(prefix, i, j)] = fold_conv_bn(ck, idx(prefix + ".conv_blocks", i, j), idx(prefix + ".bn_blocks", i, j), sp.mlp[i][j], last, 2);
            last = sp.mlp[i][j];
        }
    }
}

void fold_stack(Ckpt& ck, Folded& F, const std::string& prefix, const int* mlp, int n, int fan_in, int tail_dims) {
    int last = fan_in;
    for (int k = 0; k < n; ++k) {
        F[idx(prefix, k)] = fold_conv_bn(ck, idx(prefix + ".mlp_convs", k), idx(prefix + ".mlp_bns", k), mlp[k], last, tail_dims);
        last = mlp[k];
    }
}

Folded fold_checkpoint(Ckpt& ck, int in_channels) {
    Folded F;
    fold_msg(ck, F, "sa1", SA1, in_channels + 3);
    fold_msg(ck, F, "sa2", SA2, 320 + 3);
    fold_stack(ck, F, "sa3", SA3_MLP, 3, 512 + 3, 2);
    fold_stack(ck, F, "fp3", FP3_MLP, 2, 1536, 1);
    fold_stack(ck, F, "fp2", FP2_MLP, 2, 576, 1);
    fold_stack(ck, F, "fp1", FP1_MLP, 3, 128, 1);
    F["cls0"] = post_conv(ck, "classifier.0", "classifier.2", 256, 256, 1);
    F["cls4"] = plain_conv(ck, "classifier.4", N_CLASSES, 256, 1);
    for (int h = 0; h < 2; ++h) {
        const std::string q = std::string(SIDES[h]) + "_query_conv", p = std::string(SIDES[h]) + "_mano_regressor";
        F[q + ".0"] = post_conv(ck, q + ".0", q + ".2", 256, 256, 1, 3);
        F[q + ".4"] = fold_conv_bn(ck, q + ".4", q + ".5", 256, 256, 1, 3);
        fold_msg(ck, F, p + ".sa1", MANO_SA1, 4 + 3);
        fold_stack(ck, F, p + ".sa2", MANO_SA2_MLP, 2, 512 + 3, 2);
        F[p + ".head0"] = post_conv(ck, p + ".mano_regressor.0", p + ".mano_regressor.2", 1024, 512, 0);
        int n_out = (int)ck.dim0("left_mano_regressor.mano_regressor.4.weight");        // both hands: the same n_pose_params (TEHNet.py:144-145)
        if (n_out < MANO_OUT_MIN || n_out > MANO_OUT_MAX) n_out = 3 + 6 + 10 + 3;         // (absent / out of range: get() names the key and the expected default)
        F[p + ".head4"] = plain_conv(ck, p + ".mano_regressor.4", n_out, 1024, 0);
    }
    return F;
}

// ---------------------------------------------------------------------------------------------------- channel equalisation
// The hidden tensors of the path as (name, producers in channel order, consumers with their first input column).  Every producer
// ends in a ReLU (positively homogeneous) or a post-ReLU affine, and everything between producer and consumer (gather, max-pool,
// 3-NN interpolation, concatenation, broadcast) acts per channel.  Column offsets: pointnet2_utils.py:155,248,261,307,
// TEHNet.py:179-195.
struct Producer { std::string layer; bool post; };
struct Consumer { std::string layer; int off; };
struct Hidden { std::string name; std::vector<Producer> prod; std::vector<Consumer> cons; };

void hidden_msg(std::vector<Hidden>& T, const std::string& p, int nb, std::vector<Consumer> consumers) {
    std::vector<Producer> outs;
    for (int i = 0; i < nb; ++i) {
        T.push_back({idx(p, i) + ".h1", {{idx(p, i, 0), false}}, {{idx(p, i, 1), 0}}});
        T.push_back({idx(p, i) + ".h2", {{idx(p, i, 1), false}}, {{idx(p, i, 2), 0}}});
        outs.push_back({idx(p, i, 2), false});
    }
    T.push_back({p + ".out", outs, consumers});
}

std::vector<Hidden> hidden_tensors() {
    std::vector<Hidden> T;
    hidden_msg(T, "sa1", 3, {{"sa2.0.0", 0}, {"sa2.1.0", 0}, {"fp2.0", 0}});
    hidden_msg(T, "sa2", 2, {{"sa3.0", 3}, {"fp3.0", 0}});
    T.push_back({"sa3.h1", {{"sa3.0", false}}, {{"sa3.1", 0}}});
    T.push_back({"sa3.h2", {{"sa3.1", false}}, {{"sa3.2", 0}}});
    T.push_back({"l3", {{"sa3.2", false}}, {{"fp3.0", 512}}});
    T.push_back({"fp3.h", {{"fp3.0", false}}, {{"fp3.1", 0}}});
    T.push_back({"fp3.out", {{"fp3.1", false}}, {{"fp2.0", 320}}});
    T.push_back({"fp2.h", {{"fp2.0", false}}, {{"fp2.1", 0}}});
    T.push_back({"fp2.out", {{"fp2.1", false}}, {{"fp1.0", 0}}});
    T.push_back({"fp1.h1", {{"fp1.0", false}}, {{"fp1.1", 0}}});
    T.push_back({"fp1.h2", {{"fp1.1", false}}, {{"fp1.2", 0}}});
    // l0 also is the attention's `value` (TEHNet.py:20-26: context = sim @ value, linear in value): that consumer divides by e_c
    // through ev2h_weights.l0_unscale
    T.push_back({"l0", {{"fp1.2", false}}, {{"cls0", 0}, {"left_query_conv.0", 0}, {"right_query_conv.0", 0}}});
    T.push_back({"cls.h", {{"cls0", true}}, {{"cls4", 0}}});
    for (int h = 0; h < 2; ++h) {
        const std::string q = std::string(SIDES[h]) + "_query_conv", p = std::string(SIDES[h]) + "_mano_regressor";
        T.push_back({q + ".h", {{q + ".0", true}}, {{q + ".4", 0}}});
        hidden_msg(T, p + ".sa1", 2, {{p + ".sa2.0", 3}});
        T.push_back({p + ".sa2.h", {{p + ".sa2.0", false}}, {{p + ".sa2.1", 0}}});
        T.push_back({p + ".sa2.out", {{p + ".sa2.1", false}}, {{p + ".head0", 0}}});
        T.push_back({p + ".fc1", {{p + ".head0", true}}, {{p + ".head4", 0}}});
    }
    return T;
}

// layers whose contraction mixes hidden channels with RAW inputs (the group-all set abstractions read [xyz | features],
// pointnet2_utils.py:155): the raw columns are 0..2
bool has_raw_columns(const std::string& layer) {
    return layer == "sa3.0" || layer == "left_mano_regressor.sa2.0" || layer == "right_mano_regressor.sa2.0";
}

double median(Vec v) {
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    return (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0;
}

// sqrt of the column sums of squares of W[:, off:off+n] (all taps), rows added in order
Vec column_norms(const Layer& L, int off, int n) {
    Vec acc(n, 0.0);
    for (int o = 0; o < L.O; ++o)
        for (int c = 0; c < n; ++c) {
            const double* w = &L.W[((size_t)o * L.I + off + c) * L.T];
            double s = w[0] * w[0];
            for (int t = 1; t < L.T; ++t) s += w[t] * w[t];
            acc[c] += s;
        }
    for (int c = 0; c < n; ++c) acc[c] = std::sqrt(acc[c]);
    return acc;
}

typedef std::vector<std::pair<std::string, Vec>> Equalization;

Equalization equalize_channels(Folded& F, int sweeps = 3) {
    Equalization acc;
    const std::vector<Hidden> tensors = hidden_tensors();
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        for (size_t ti = 0; ti < tensors.size(); ++ti) {
            const Hidden& H = tensors[ti];
            Vec r;
            for (const Producer& pr : H.prod) {
                const Layer& L = F.at(pr.layer);
                const size_t row = (size_t)L.I * L.T;
                for (int o = 0; o < L.O; ++o) {
                    double s = 0.0;
                    for (size_t k = 0; k < row; ++k) s += L.W[o * row + k] * L.W[o * row + k];
                    s += L.b[o] * L.b[o];
                    const double rn = std::sqrt(s);
                    r.push_back(pr.post ? std::fabs(L.ps[o]) * rn + std::fabs(L.pt[o]) : rn);
                }
            }
            const int n = (int)r.size();
            Vec om(n, 0.0);
            for (const Consumer& cs : H.cons) {
                const Vec cn = column_norms(F.at(cs.layer), cs.off, n);
                double ss = 0.0;
                for (int c = 0; c < n; ++c) ss += cn[c] * cn[c];
                const double rms = std::sqrt(ss / n);
                if (rms > 0)
                    for (int c = 0; c < n; ++c) { const double q = cn[c] / rms; om[c] += q * q; }
            }
            std::vector<char> ok(n);
            Vec lg(n, 0.0), lgok;
            for (int c = 0; c < n; ++c) {
                om[c] = std::sqrt(om[c]);
                ok[c] = r[c] > 0 && om[c] > 0 && std::isfinite(r[c]) && std::isfinite(om[c]);
                if (ok[c]) { lg[c] = 0.5 * (std::log2(om[c]) - std::log2(r[c])); lgok.push_back(lg[c]); }
            }
            const bool any = !lgok.empty();
            if (any) {
                const double med = median(lgok);          // keep the tensor's overall magnitude where the checkpoint put it ...
                for (int c = 0; c < n; ++c) if (ok[c]) lg[c] -= med;
            }
            for (const Consumer& cs : H.cons) {           // ... unless a consumer mixes it with raw inputs in one contraction:
                if (!has_raw_columns(cs.layer) || !any) continue;      // then the hidden columns are brought level with the raw ones
                const Layer& L = F.at(cs.layer);
                const Vec raw = column_norms(L, 0, 3), hid_all = column_norms(L, cs.off, n);
                Vec hid;
                for (int c = 0; c < n; ++c) if (ok[c]) hid.push_back(hid_all[c] / std::exp2(std::nearbyint(lg[c])));
                const double mr = median(raw), mh = median(hid);
                if (mr > 0 && mh > 0) {
                    const double shift = std::nearbyint(std::log2(mr / mh));
                    for (int c = 0; c < n; ++c) if (ok[c]) lg[c] -= shift;
                }
            }
            Vec e(n);
            for (int c = 0; c < n; ++c) e[c] = std::exp2(std::min(40.0, std::max(-40.0, std::nearbyint(lg[c]))));
            int o0 = 0;
            for (const Producer& pr : H.prod) {
                Layer& L = F.at(pr.layer);
                const size_t row = (size_t)L.I * L.T;
                for (int o = 0; o < L.O; ++o) {
                    const double ek = e[o0 + o];
                    if (pr.post) { L.ps[o] = L.ps[o] * ek; L.pt[o] = L.pt[o] * ek; }
                    else {
                        for (size_t k = 0; k < row; ++k) L.W[o * row + k] = L.W[o * row + k] * ek;
                        L.b[o] = L.b[o] * ek;
                    }
                }
                o0 += L.O;
            }
            for (const Consumer& cs : H.cons) {
                Layer& L = F.at(cs.layer);
                for (int o = 0; o < L.O; ++o)
                    for (int c = 0; c < n; ++c)
                        for (int t = 0; t < L.T; ++t) {
                            double& w = L.W[((size_t)o * L.I + cs.off + c) * L.T + t];
                            w = w / e[c];
                        }
            }
            if (sweep == 0) acc.push_back({H.name, Vec(n, 1.0)});
            Vec& a = acc[ti].second;
            for (int c = 0; c < n; ++c) a[c] = a[c] * e[c];
        }
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------------- operand planes
// fp32 -> 16-bit plane patterns, the same splits the kernels apply to activations (csrc/planes.hpp: split_planes)
inline uint16_t f16_bits(float x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }      // RNE, overflow -> inf
inline float f16_value(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

// mode code -> planes stored per operand / fp16 planes with the W / u range factor (csrc/planes.hpp: plane_count, planes_f16)
inline int npl(int ns) { return ns == 4 ? 1 : ns; }
inline bool is_f16(int ns) { return ns == 2 || ns == 4; }

void split_planes_host(const std::vector<float>& x, int ns, std::vector<uint16_t>* planes) {
    const size_t n = x.size();
    for (int s = 0; s < npl(ns); ++s) planes[s].resize(n);
    if (is_f16(ns)) {
        for (size_t i = 0; i < n; ++i)
            if (std::fabs(x[i]) >= 65504.0f) fail("f16x2 / f16 need |weight| < 65504 (fp16 range); use precision bf16x3 or f32 for this checkpoint");
        for (size_t i = 0; i < n; ++i) {
            const uint16_t h = f16_bits(x[i]);
            planes[0][i] = h;
            if (ns == 2) planes[1][i] = f16_bits(x[i] - f16_value(h));
        }
    } else if (ns == 1) {
        for (size_t i = 0; i < n; ++i) {
            uint32_t u; memcpy(&u, &x[i], 4);
            const uint64_t w = (uint64_t)u + 0x7FFF + ((u >> 16) & 1);
            planes[0][i] = (uint16_t)(w >> 16);
        }
    } else {
        for (size_t i = 0; i < n; ++i) {
            float r = x[i];
            for (int s = 0; s < ns; ++s) {
                uint32_t u; memcpy(&u, &r, 4);
                u &= 0xFFFF0000u;
                planes[s][i] = (uint16_t)(u >> 16);
                float t; memcpy(&t, &u, 4);
                r = r - t;                                  // exact: the difference has fewer significant bits
            }
        }
    }
}

// Power of two u such that the planes are taken of W / u.  F16X2 / F16 only: fp16 has 5 exponent bits, so the low plane of a weight below
// 2^-3 is subnormal and small-magnitude layers lose accuracy (measured 2.6e-4 at |W| ~ 1e-4).  Dividing by u = 2^-k with
// max|W / u| in [2^13, 2^14) is exact and the kernels multiply the accumulated product by u (also exact).
double plane_unscale(const double* W, size_t count, int ns) {
    if (!is_f16(ns)) return 1.0;
    double m = 0.0;
    for (size_t i = 0; i < count; ++i) { const double a = std::fabs(W[i]); if (a > m || a != a) m = a; }
    if (m == 0.0 || !std::isfinite(m)) return 1.0;
    int ex;
    (void)std::frexp(16384.0 / m, &ex);                     // 16384 / m = f * 2^ex, f in [0.5, 1): floor(log2) = ex - 1
    const int k = std::max(-24, std::min(ex - 1, 60));
    return std::ldexp(1.0, -k);
}

// F16 (one fp16 plane) [r6]: the factor of a chain's SECOND matrix.  The fused set-abstraction kernels keep ONE power of two per window
// for the whole chain in this mode (layer 2's accumulators are converted as they are: no factor between the layers, csrc/sa_mlp_bf16.hip
// C2ONE), so H2' = (s1 / u2) H2 must land in the fp16 range next to H1' = s1 H1: u2 = 2^floor(log2 |W2|_1) (largest row L1 norm) makes
// the two bounds agree to a factor of 2-4.  One plane has no low part to protect: a weight 2^-14 below the row norm is still normal.
double chain_unscale(const double* W, int rows, int cols) {
    double m = 0.0;
    for (int r = 0; r < rows; ++r) {
        double a = 0.0;
        for (int k = 0; k < cols; ++k) a += std::fabs(W[(size_t)r * cols + k]);
        if (a > m || a != a) m = a;
    }
    if (m == 0.0 || !std::isfinite(m)) return 1.0;
    int ex;
    (void)std::frexp(m, &ex);                               // m = f * 2^ex, f in [0.5, 1): floor(log2 m) = ex - 1
    return std::ldexp(1.0, std::max(-60, std::min(ex - 1, 24)));
}

void sa_geometry(int C2, int* T2, int* C2P) {
    *T2 = up(C2, 32) / 32;
    const int rem = C2 % 32, m_last = rem == 0 ? 2 : (rem <= 16 ? 1 : 2);
    *C2P = 32 * (*T2 - 1) + 16 * m_last;
}

struct SaImages { std::vector<uint8_t> i2, i3; float u2 = 1.f, u3 = 1.f; };

// Byte images of the LDS weight tiles of sa_mlp_max_bf16_kernel (SaBCfg in csrc/sa_mlp_bf16.hip).  W2 [C2][C1], W3 [C3][C2].
SaImages sa_images(const double* W2in, const double* W3in, int C1, int C2, int C3, int ns) {
    SaImages R;
    const double u2 = ns == 4 ? chain_unscale(W2in, C2, C1) : plane_unscale(W2in, (size_t)C2 * C1, ns), u3 = plane_unscale(W3in, (size_t)C3 * C2, ns);
    R.u2 = (float)u2; R.u3 = (float)u3;
    int T2, C2P;
    sa_geometry(C2, &T2, &C2P);
    int g[10];
    if (ev2h_tile_geometry(C1, C2, C3, ns, g) != EV2H_OK) fail("unsupported chain %d-%d-%d", C1, C2, C3);
    const int left = g[8];            // 0, or the 1..4 channels past the last full tile whose plane products share MFMAs (SaBCfg::PACK4)
    const int base = 32 * (T2 - 1);
    const int R2 = T2 * 32;
    std::vector<float> W2p((size_t)R2 * C1, 0.f);
    for (int r = 0; r < C2; ++r)
        for (int k = 0; k < C1; ++k) W2p[(size_t)r * C1 + k] = (float)(W2in[(size_t)r * C1 + k] / u2);
    if (g[9]) {      // BF16: k slots in the D-register order of the layer-1 MFMA (ev2h_tile_geometry W2PERM)
        std::vector<float> q(W2p.size());
        for (int r = 0; r < R2; ++r)
            for (int c = 0; c < C1 / 32; ++c)
                for (int h = 0; h < 2; ++h)
                    for (int m = 0; m < 2; ++m)
                        for (int e = 0; e < 8; ++e)
                            q[(size_t)r * C1 + 32 * c + 16 * h + 8 * m + e] = W2p[(size_t)r * C1 + 32 * c + 16 * m + 4 * h + (e & 3) + 8 * (e >> 2)];
        W2p.swap(q);
    }
    std::vector<uint16_t> p2[3];
    split_planes_host(W2p, ns, p2);
    if (left) {
        if (!(ns == 2 && C2 - base == left && left <= 4)) fail("leftover-channel packing: inconsistent geometry");
        for (int r = 0; r < left; ++r)      // rows 8.. of the high-plane image: the leftover rows' LOW plane
            memcpy(&p2[0][(size_t)(base + 8 + r) * C1], &p2[1][(size_t)(base + r) * C1], (size_t)C1 * 2);
    }
    const int rs2 = npl(ns) * 64 + 16;
    R.i2.assign((size_t)(C1 / 32) * R2 * rs2, 0);
    for (int c = 0; c < C1 / 32; ++c)
        for (int s = 0; s < npl(ns); ++s)
            for (int r = 0; r < R2; ++r)
                memcpy(&R.i2[((size_t)c * R2 + r) * rs2 + s * 64], &p2[s][(size_t)r * C1 + 32 * c], 64);
    // layer-3 contraction order follows the MFMA D layout of layer 2: position 32t+16m+8h+e <-> channel 32t+16m+4h+(e&3)+8(e>>2)
    std::vector<float> W3p((size_t)C3 * C2P, 0.f);
    for (int pos = 0; pos < C2P; ++pos) {
        const int t = pos / 32, w = pos % 32, m = w / 16, h = (w % 16) / 8, e = w % 8;
        const int ch = 32 * t + 16 * m + 4 * h + (e & 3) + 8 * (e >> 2);
        if (ch < C2)
            for (int o = 0; o < C3; ++o) W3p[(size_t)o * C2P + pos] = (float)(W3in[(size_t)o * C2 + ch] / u3);
    }
    std::vector<uint16_t> p3[3];
    split_planes_host(W3p, ns, p3);
    if (left) {                                           // last 16 k-slots of every row: [wh | wh | wl | 0]
        for (int o = 0; o < C3; ++o) {
            uint16_t* hrow = &p3[0][(size_t)o * C2P + base];
            const uint16_t* lrow = &p3[1][(size_t)o * C2P + base];
            for (int j = 0; j < 4; ++j) { hrow[4 + j] = hrow[j]; hrow[8 + j] = lrow[j]; hrow[12 + j] = 0; }
        }
    }
    const int rs3 = npl(ns) * C2P * 2 + 16;
    R.i3.assign((size_t)(C3 / 32) * 32 * rs3, 0);
    for (int s = 0; s < npl(ns); ++s)
        for (int o = 0; o < C3; ++o)
            memcpy(&R.i3[(size_t)o * rs3 + (size_t)s * C2P * 2], &p3[s][(size_t)o * C2P], (size_t)C2P * 2);
    if (g[0] != T2 || g[1] != C2P || g[2] != rs2 || g[3] != rs3 || g[4] != R2 * rs2 || g[5] != 32 * rs3)
        fail("tile geometry of chain %d-%d-%d (planes %d): the packer builds T2=%d C2P=%d RS2=%d RS3=%d, the kernels expect T2=%d C2P=%d RS2=%d RS3=%d "
             "TB2=%d TB3=%d", C1, C2, C3, ns, T2, C2P, rs2, rs3, g[0], g[1], g[2], g[3], g[4], g[5]);
    return R;
}

// Plane images of a dense weight W [N][K] for the 16-bit GEMM kernels: for every `rows`-row N tile and every 32-wide K tile one
// LDS tile image [rows][ns*64 + 16 bytes] (planes side by side, 16 B row pad)
std::vector<uint8_t> gemm_image(const double* W, int N, int K, int ns, int rows, float* u_out) {
    const double u = plane_unscale(W, (size_t)N * K, ns);
    *u_out = (float)u;
    const int tn = up(N, rows) / rows, nk = up(K, 32) / 32;
    const int Np = tn * rows, Kp = nk * 32;
    std::vector<float> Wp((size_t)Np * Kp, 0.f);
    for (int r = 0; r < N; ++r)
        for (int k = 0; k < K; ++k) Wp[(size_t)r * Kp + k] = (float)(W[(size_t)r * K + k] / u);
    std::vector<uint16_t> pl[3];
    split_planes_host(Wp, ns, pl);
    const int rs = npl(ns) * 64 + 16;
    int g[10];
    if (ev2h_tile_geometry(128, 128, 256, ns, g) != EV2H_OK || g[6] != rs || g[7] != 32)
        fail("dense W image geometry: the packer builds rows of %d B x 32 k, the kernels expect %d B x %d k", rs, g[6], g[7]);
    std::vector<uint8_t> img((size_t)tn * nk * rows * rs, 0);
    for (int s = 0; s < npl(ns); ++s)
        for (int a = 0; a < tn; ++a)
            for (int kk = 0; kk < nk; ++kk)
                for (int r = 0; r < rows; ++r)
                    memcpy(&img[(((size_t)a * nk + kk) * rows + r) * rs + s * 64], &pl[s][(size_t)(a * rows + r) * Kp + 32 * kk], 64);
    return img;
}

// ---------------------------------------------------------------------------------------------------- the packed object
struct Blob { std::string name; int dtype; int rows, cols; size_t off, bytes; };
// F16X2: how a matrix sits inside its ONE power-of-two scale: weights with 0 < |w / u| < 2^-3 (more than ~2^17 below the matrix
// maximum: low plane subnormal) and < 2^-14 (high plane subnormal too)
struct WSpread { std::string name; uint64_t nonzero = 0, low = 0, high = 0; };

WSpread weight_spread(const std::string& name, const double* W, size_t count, double u) {
    WSpread r;
    r.name = name;
    for (size_t i = 0; i < count; ++i) {
        const double a = std::fabs(W[i] / u);
        if (a > 0) { ++r.nonzero; r.low += a < 0.125; r.high += a < 6.103515625e-05; }
    }
    return r;
}

}  // namespace

struct ev2h_packed {
    ev2h_weights w;
    std::vector<char> host;
    std::vector<Blob> blobs;
    void* dev = nullptr;
    size_t dev_bytes = 0;
    Equalization eq;
    std::vector<WSpread> wspread;
};

namespace {

constexpr int GEMM_W_TILE_ROWS = 128;      // rows per W image tile (128: occupancy kernel, 256: wide kernel)

struct Builder {
    ev2h_packed& P;
    int ns;
    // plane-mode code of one kernel family: the F16 mode packs the families of ev2h_weights.f16_families with one fp16 plane (4)
    // and the others as F16X2 (2); every other precision has one code for everything
    int nsf(int fam) const { return P.w.precision == EV2H_PREC_F16 ? ((P.w.f16_families & fam) ? 4 : 2) : ns; }
    std::vector<std::pair<const void**, size_t>> fix;     // pointer fields of P.w and the arena offsets they get

    size_t put(const std::string& name, int dtype, int rows, int cols, const void* data, size_t bytes) {
        const size_t off = (P.host.size() + 255) / 256 * 256;
        P.host.resize(off + bytes, 0);
        if (bytes) memcpy(&P.host[off], data, bytes);
        P.blobs.push_back({name, dtype, rows, cols, off, bytes});
        return off;
    }
    template <class T> void dev(T const** field, const std::string& name, const double* a, int rows, int cols) {
        std::vector<float> f((size_t)rows * cols);
        for (size_t i = 0; i < f.size(); ++i) f[i] = (float)a[i];
        fix.push_back({(const void**)field, put(name, EV2H_DT_F32, rows, cols, f.data(), f.size() * 4)});
    }
    template <class T> void vec(T const** field, const std::string& name, const Vec& a) {
        std::vector<float> f(a.size());
        for (size_t i = 0; i < f.size(); ++i) f[i] = (float)a[i];
        fix.push_back({(const void**)field, put(name, EV2H_DT_F32, 0, (int)a.size(), f.data(), f.size() * 4)});
    }
    template <class T> void dev_bytes(T const** field, const std::string& name, const std::vector<uint8_t>& img) {
        fix.push_back({(const void**)field, put(name, -1, 0, (int)img.size(), img.data(), img.size())});
    }

    static Vec pad2(const double* a, int r0, int c0, int lda, int rows, int cols) {
        Vec out((size_t)rows * cols, 0.0);
        for (int r = 0; r < r0; ++r)
            for (int c = 0; c < c0; ++c) out[(size_t)r * cols + c] = a[(size_t)r * lda + c];
        return out;
    }
    static float bound(double m) { return (float)(m * (1 + 1e-6)); }     // rounded up a little: fp32 rounding can never make a bound too small
    static double max_row_l1(const double* W, int rows, int cols, int ld) {
        double m = 0.0;
        for (int r = 0; r < rows; ++r) {
            double s = 0.0;
            for (int c = 0; c < cols; ++c) s += std::fabs(W[(size_t)r * ld + c]);
            m = std::max(m, s);
        }
        return m;
    }
    static double max_abs(const Vec& v) { double m = 0.0; for (double x : v) m = std::max(m, std::fabs(x)); return m; }

    // one folded Conv/Linear as the GEMM kernels take it: W [O][Kfull] (row-major), K per tap (0 = Kfull padded)
    void dense(ev2h_dense& d, const std::string& name, const Vec& W, int O, int Kfull, const Vec* b, const Vec* ps = nullptr, const Vec* pt = nullptr,
               int K = 0, int fam = EV2H_FAM_DENSE) {
        const int ns = nsf(fam);
        const int ldw = up(Kfull, 4);
        const Vec Wp = pad2(W.data(), O, Kfull, Kfull, O, ldw);
        dev(&d.W, name + ".W", Wp.data(), O, ldw);
        if (b) vec(&d.b, name + ".b", *b);
        if (ps) vec(&d.post_scale, name + ".ps", *ps);
        if (pt) vec(&d.post_shift, name + ".pt", *pt);
        d.O = O; d.K = K ? K : ldw; d.ldw = ldw;
        if (is_f16(ns)) P.wspread.push_back(weight_spread(name, W.data(), W.size(), plane_unscale(W.data(), W.size(), ns)));
        if (ns && O >= 96) {             // all but the tiny heads: pre-split W images, streamed by LDS-DMA
            dev_bytes(&d.Ws, name + ".Ws", gemm_image(Wp.data(), O, ldw, ns, GEMM_W_TILE_ROWS, &d.w_unscale));
            d.ws_tile_rows = GEMM_W_TILE_ROWS;
        } else {
            d.w_unscale = (float)plane_unscale(W.data(), W.size(), ns);        // the kernel splits W / w_unscale on the fly
        }
    }

    // sample_and_group_all concatenates [xyz(3), features(512)] (pointnet2_utils.py:155); our buffers hold [features(512) | xyz(3) |
    // 0 x 5] so that K = 520 is a multiple of 8
    void group_all(ev2h_dense* arr, const Folded& F, const std::string& prefix, int nlayers) {
        for (int k = 0; k < nlayers; ++k) {
            const Layer& L = F.at(idx(prefix, k));
            if (k == 0) {
                if (L.I != 515) fail("group-all input width %d != 515", L.I);
                Vec W((size_t)L.O * 520, 0.0);
                for (int o = 0; o < L.O; ++o) {
                    for (int c = 0; c < 512; ++c) W[(size_t)o * 520 + c] = L.W[(size_t)o * 515 + 3 + c];
                    for (int c = 0; c < 3; ++c) W[(size_t)o * 520 + 512 + c] = L.W[(size_t)o * 515 + c];
                }
                dense(arr[k], idx(prefix, k), W, L.O, 520, &L.b);
            } else {
                dense(arr[k], idx(prefix, k), L.W, L.O, L.I, &L.b);
            }
        }
    }

    void chain_images(ev2h_sa_branch& br, const std::string& n, const double* W2, const double* W3, int C1, int C2, int C3, int fam) {
        const int ns = nsf(fam);
        SaImages im = sa_images(W2, W3, C1, C2, C3, ns);
        br.w2_unscale = im.u2; br.w3_unscale = im.u3;
        if (is_f16(ns)) {
            P.wspread.push_back(weight_spread(n + ".W2", W2, (size_t)C2 * C1, plane_unscale(W2, (size_t)C2 * C1, ns)));      // (F16: u2 is the chain's factor, the spread is still counted against the largest weight)
            P.wspread.push_back(weight_spread(n + ".W3", W3, (size_t)C3 * C2, im.u3));
        }
        dev_bytes(&br.W2s, n + ".W2s", im.i2);
        dev_bytes(&br.W3s, n + ".W3s", im.i3);
    }

    void sa_module(ev2h_sa_module& m, const Folded& F, const std::string& prefix, const MsgSpec& sp, int nfeat, int kf) {
        const int ns = nsf(EV2H_FAM_SA);
        m.kf = kf; m.npoint = sp.npoint; m.nbranch = sp.nb;
        int c1sum = 0;
        for (int i = 0; i < sp.nb; ++i) c1sum += sp.mlp[i][0];
        Vec W1f((size_t)c1sum * kf, 0.0), b1;
        int r0 = 0;
        for (int i = 0; i < sp.nb; ++i) {
            const Layer &L0 = F.at(idx(prefix, i, 0)), &L1 = F.at(idx(prefix, i, 1)), &L2 = F.at(idx(prefix, i, 2));
            const int C1 = L0.O, C2 = L1.O, C3 = L2.O;
            if (L0.I != nfeat + 3) fail("%s: layer-1 fan-in %d != %d", prefix.c_str(), L0.I, nfeat + 3);      // [features..., dx, dy, dz] (pointnet2_utils.py:248)
            for (int r = 0; r < C1; ++r)
                for (int c = 0; c < nfeat; ++c) W1f[(size_t)(r0 + r) * kf + c] = L0.W[(size_t)r * L0.I + c];
            b1.insert(b1.end(), L0.b.begin(), L0.b.end());
            ev2h_sa_branch& br = m.br[i];
            const std::string n = idx(prefix, i);
            const Vec W1x = pad2(&L0.W[nfeat], C1, 3, L0.I, C1, 4);
            dev(&br.W1x, n + ".W1x", W1x.data(), C1, 4);
            const Vec W2p = pad2(L1.W.data(), C2, C1, C1, up(C2, 32), C1);
            dev(&br.W2, n + ".W2", W2p.data(), up(C2, 32), C1);
            Vec b2p(up(C2, 32), 0.0);
            std::copy(L1.b.begin(), L1.b.end(), b2p.begin());
            vec(&br.b2, n + ".b2", b2p);
            const Vec W3p = pad2(L2.W.data(), C3, C2, C2, C3, up(C2, 8));
            dev(&br.W3, n + ".W3", W3p.data(), C3, up(C2, 8));
            vec(&br.b3, n + ".b3", L2.b);
            br.C1 = C1; br.C2 = C2; br.C3 = C3; br.K = sp.nsample[i]; br.radius = sp.radius[i];
            br.w1x_norm = bound(max_row_l1(&L0.W[nfeat], C1, 3, L0.I));
            {   // plane factors of the two column blocks of this branch's layer 1 (the layer-1 MFMA of the F16X2 feature mode)
                Vec wf((size_t)C1 * nfeat);
                for (int r = 0; r < C1; ++r)
                    for (int c = 0; c < nfeat; ++c) wf[(size_t)r * nfeat + c] = L0.W[(size_t)r * L0.I + c];
                br.w1f_unscale = (float)plane_unscale(wf.data(), wf.size(), ns);
                br.w1x_unscale = (float)plane_unscale(W1x.data(), W1x.size(), ns);
            }
            br.w2_norm = bound(max_row_l1(L1.W.data(), C2, C1, C1));
            br.b2_max = bound(max_abs(L1.b));
            br.w3_norm = bound(max_row_l1(L2.W.data(), C3, C2, C2));
            br.b3_max = bound(max_abs(L2.b));
            if (ns) chain_images(br, n, L1.W.data(), L2.W.data(), C1, C2, C3, EV2H_FAM_SA);
            r0 += C1;
        }
        dev(&m.W1f, prefix + ".W1f", W1f.data(), c1sum, kf);
        vec(&m.b1, prefix + ".b1", b1);
        if (ns && kf >= 32) {            // enc.sa2 (K = 320): plane images, the fast GEMM kernel
            dev_bytes(&m.W1fs, prefix + ".W1fs", gemm_image(W1f.data(), c1sum, kf, ns, GEMM_W_TILE_ROWS, &m.w1f_unscale));
            if (is_f16(ns)) P.wspread.push_back(weight_spread(prefix + ".W1f", W1f.data(), W1f.size(), m.w1f_unscale));
        } else {
            m.w1f_unscale = (float)plane_unscale(W1f.data(), W1f.size(), ns);      // K = 8 tables run as fp32 fma chains (table_k8_kernel)
        }
        m.w1f_norm = bound(max_row_l1(W1f.data(), c1sum, kf, kf));
        m.b1_max = bound(max_abs(b1));
    }

    // A three-layer feature-propagation MLP without skip input (fp1, TEHNet.py:129) in the form ev2h_fp_mlp takes: the first layer
    // as a table over the coarse points (W1f, b1 -- it commutes with the interpolation), layers 2-3 as the tile images of the fused
    // set-abstraction kernel
    void fp_module(ev2h_sa_module& m, const Folded& F, const std::string& prefix) {
        const int ns = nsf(EV2H_FAM_ROWS);
        const Layer &L0 = F.at(idx(prefix, 0)), &L1 = F.at(idx(prefix, 1)), &L2 = F.at(idx(prefix, 2));
        m.kf = L0.I; m.npoint = 0; m.nbranch = 1;
        if (!(L0.O == 128 && L1.O == 128 && L2.O == 256 && m.kf % 32 == 0)) fail("fp1 chain is not 128-128-256");
        ev2h_sa_branch& br = m.br[0];
        const std::string n = prefix + "m";
        vec(&br.b2, n + ".b2", L1.b);
        vec(&br.b3, n + ".b3", L2.b);
        br.C1 = L0.O; br.C2 = L1.O; br.C3 = L2.O; br.K = 32; br.radius = 0.0;
        br.w1x_norm = 0.f;
        br.w2_norm = bound(max_row_l1(L1.W.data(), L1.O, L1.I, L1.I));
        br.b2_max = bound(max_abs(L1.b));
        br.w3_norm = bound(max_row_l1(L2.W.data(), L2.O, L2.I, L2.I));
        br.b3_max = bound(max_abs(L2.b));
        chain_images(br, n, L1.W.data(), L2.W.data(), L0.O, L1.O, L2.O, EV2H_FAM_ROWS);
        dev(&m.W1f, n + ".W1f", L0.W.data(), L0.O, L0.I);
        vec(&m.b1, n + ".b1", L0.b);
        dev_bytes(&m.W1fs, n + ".W1fs", gemm_image(L0.W.data(), L0.O, L0.I, ns, GEMM_W_TILE_ROWS, &m.w1f_unscale));
        if (is_f16(ns)) P.wspread.push_back(weight_spread(n + ".W1f", L0.W.data(), L0.W.size(), m.w1f_unscale));
        m.w1f_norm = bound(max_row_l1(L0.W.data(), L0.O, L0.I, L0.I));
        m.b1_max = bound(max_abs(L0.b));
    }

    // The segmentation head Conv1d -> ReLU -> BN -> (Dropout) -> Conv1d (TEHNet.py:135-141) as a two-layer ev2h_fp_mlp chain: the BN
    // affine y = alpha relu(z) + beta sits between a ReLU and a k=1 convolution, so it folds forward exactly (float64, rounded
    // once): W4' = W4 diag(alpha), b4' = W4 beta + b4.  The 4 output rows are zero-padded to one 32-row tile.
    void cls_module(ev2h_sa_branch& br, const Layer& c0, const Layer& c4) {
        const int C2 = c0.O, C1 = c0.I;
        if (!(C1 == 256 && C2 == 256 && c4.O <= 32)) fail("segmentation head is not 256-256-<=32");
        Vec W4p((size_t)32 * C2, 0.0), b4p(32, 0.0);
        for (int o = 0; o < c4.O; ++o) {
            double s = 0.0;                                   // W4 @ beta in k order, then + b4
            for (int k = 0; k < C2; ++k) {
                W4p[(size_t)o * C2 + k] = c4.W[(size_t)o * C2 + k] * c0.ps[k];
                s += c4.W[(size_t)o * C2 + k] * c0.pt[k];
            }
            b4p[o] = s + c4.b[o];
        }
        vec(&br.b2, "clsm.b2", c0.b);
        vec(&br.b3, "clsm.b3", b4p);
        br.C1 = C1; br.C2 = C2; br.C3 = 32; br.K = 32; br.radius = 0.0;
        br.w1x_norm = 0.f;
        br.w2_norm = bound(max_row_l1(c0.W.data(), C2, C1, C1));
        br.b2_max = bound(max_abs(c0.b));
        br.w3_norm = bound(max_row_l1(W4p.data(), 32, C2, C2));
        br.b3_max = bound(max_abs(b4p));
        chain_images(br, "clsm", c0.W.data(), W4p.data(), C1, C2, 32, EV2H_FAM_ROWS);
    }

    void build(Folded& F, int in_channels) {
        ev2h_weights& w = P.w;
        sa_module(w.sa1, F, "sa1", SA1, in_channels, 8);
        sa_module(w.sa2, F, "sa2", SA2, 320, 320);
        for (int h = 0; h < 2; ++h) {
            const std::string p = std::string(SIDES[h]) + "_mano_regressor";
            sa_module(w.mano_sa1[h], F, p + ".sa1", MANO_SA1, 4, 8);
            group_all(w.mano_sa2[h], F, p + ".sa2", 2);
            const Layer &L0 = F.at(p + ".head0"), &L4 = F.at(p + ".head4");
            dense(w.head0[h], p + ".head0", L0.W, L0.O, L0.I, &L0.b, &L0.ps, &L0.pt);
            dense(w.head4[h], p + ".head4", L4.W, L4.O, L4.I, &L4.b);
        }
        group_all(w.sa3, F, "sa3", 3);
        {   // fp3: 1536 = 512 skip (l2_points) + 1024 broadcast (l3_points), pointnet2_utils.py:293-294,307
            const Layer& L = F.at("fp3.0");
            Vec Ws((size_t)L.O * 512), Wb((size_t)L.O * 1024);
            for (int o = 0; o < L.O; ++o) {
                std::copy(&L.W[(size_t)o * 1536], &L.W[(size_t)o * 1536 + 512], &Ws[(size_t)o * 512]);
                std::copy(&L.W[(size_t)o * 1536 + 512], &L.W[(size_t)o * 1536 + 1536], &Wb[(size_t)o * 1024]);
            }
            dense(w.fp3_skip, "fp3.skip", Ws, L.O, 512, nullptr);
            dense(w.fp3_bcast, "fp3.bcast", Wb, L.O, 1024, &L.b);
            const Layer& L1 = F.at("fp3.1");
            dense(w.fp3_1, "fp3.1", L1.W, L1.O, L1.I, &L1.b);
        }
        for (int k = 0; k < 2; ++k) { const Layer& L = F.at(idx("fp2", k)); dense(w.fp2[k], idx("fp2", k), L.W, L.O, L.I, &L.b); }
        for (int k = 0; k < 3; ++k) { const Layer& L = F.at(idx("fp1", k)); dense(w.fp1[k], idx("fp1", k), L.W, L.O, L.I, &L.b); }
        if (ns) fp_module(w.fp1m, F, "fp1");
        const Layer &c0 = F.at("cls0"), &c4 = F.at("cls4");
        dense(w.cls0, "cls0", c0.W, c0.O, c0.I, &c0.b, &c0.ps, &c0.pt);
        dense(w.cls4, "cls4", c4.W, c4.O, c4.I, &c4.b);
        if (ns) cls_module(w.clsm, c0, c4);
        // query convs: tap-major [O][3*I]; both hands' first conv stacked along O
        Vec W0, b0, a0, be0;
        auto tap_major = [](const Layer& L) {
            Vec out((size_t)L.O * L.T * L.I);
            for (int o = 0; o < L.O; ++o)
                for (int i = 0; i < L.I; ++i)
                    for (int t = 0; t < L.T; ++t) out[((size_t)o * L.T + t) * L.I + i] = L.W[((size_t)o * L.I + i) * L.T + t];
            return out;
        };
        for (int h = 0; h < 2; ++h) {
            const std::string p = std::string(SIDES[h]) + "_query_conv";
            const Layer& L = F.at(p + ".0");
            const Vec t0 = tap_major(L);
            W0.insert(W0.end(), t0.begin(), t0.end());
            b0.insert(b0.end(), L.b.begin(), L.b.end());
            a0.insert(a0.end(), L.ps.begin(), L.ps.end());
            be0.insert(be0.end(), L.pt.begin(), L.pt.end());
            const Layer& L4 = F.at(p + ".4");
            const Vec W4 = tap_major(L4);                                          // [256][768]
            dense(w.qconv4[h], p + ".4", W4, L4.O, 3 * L4.I, &L4.b, nullptr, nullptr, 256);
            Vec W4T((size_t)768 * 256);                                            // [768][256]: ev2h_attn_sim_folded
            for (int o = 0; o < 256; ++o)
                for (int k = 0; k < 768; ++k) W4T[(size_t)k * 256 + o] = W4[(size_t)o * 768 + k];
            dev(&w.qconv4T[h], p + ".4.WT", W4T.data(), 768, 256);
        }
        dense(w.qconv0, "qconv0", W0, 512, 768, &b0, &a0, &be0, 256, EV2H_FAM_QCONV);
        // the attention's `value` is l0 itself (TEHNet.py:20-26): its channels are divided by their equalisation factor there
        for (const auto& kv : P.eq)
            if (kv.first == "l0") {
                bool any = false;
                Vec inv(kv.second.size());
                for (size_t c = 0; c < inv.size(); ++c) { inv[c] = 1.0 / kv.second[c]; any = any || kv.second[c] != 1.0; }
                if (any) vec(&w.l0_unscale, "l0.unscale", inv);
            }
    }
};

int planes_of(int precision) {
    switch (precision) {
        case EV2H_PREC_F32: return 0;
        case EV2H_PREC_BF16: return 1;
        case EV2H_PREC_F16X2: return 2;
        case EV2H_PREC_BF16X3: return 3;
        case EV2H_PREC_F16: return 4;          // mode code: one fp16 plane (npl), fp16 range factors (is_f16)
    }
    fail("unknown precision %d", precision);
}

bool ends_with(const std::string& s, const char* tail) {
    const size_t n = strlen(tail);
    return s.size() >= n && !s.compare(s.size() - n, n, tail);
}

}  // namespace

#define EV2H_TRY try {
#define EV2H_CATCH                                            \
    }                                                         \
    catch (const std::exception& e) {                         \
        ev2h_set_error("%s", e.what());                       \
        return EV2H_ERR_ARG;                                  \
    }

extern "C" int ev2h_pack_weights(const ev2h_tensor_desc* tensors, int n, int in_channels, int precision, int flags, ev2h_packed** out) {
    EV2H_CHECK_ARG(tensors && n > 0 && out && (in_channels == 4 || in_channels == 5));
    *out = nullptr;
    ev2h_packed* P = nullptr;
    try {
        const int ns = planes_of(precision);
        Ckpt ck;
        for (int i = 0; i < n; ++i) {
            const ev2h_tensor_desc& d = tensors[i];
            if (!d.name || d.ndim < 0 || d.ndim > 4) fail("ev2h_pack_weights: tensor %d: bad descriptor", i);
            std::string name = d.name;
            if (!name.compare(0, 7, "module.")) name = name.substr(7);            // nn.DataParallel checkpoints (model.py:16-21)
            if (ends_with(name, "num_batches_tracked")) continue;
            size_t cnt = 1;
            Tensor t;
            for (int k = 0; k < d.ndim; ++k) { if (d.shape[k] < 0) fail("ev2h_pack_weights: \"%s\": negative dimension", d.name); t.shape.push_back(d.shape[k]); cnt *= (size_t)d.shape[k]; }
            if (!d.data && cnt) fail("ev2h_pack_weights: \"%s\": NULL data", d.name);
            t.v.resize(cnt);
            if (d.dtype == EV2H_DT_F32) { const float* s = (const float*)d.data; for (size_t k = 0; k < cnt; ++k) t.v[k] = s[k]; }
            else if (d.dtype == EV2H_DT_F64) { const double* s = (const double*)d.data; for (size_t k = 0; k < cnt; ++k) t.v[k] = s[k]; }
            else fail("ev2h_pack_weights: \"%s\": dtype %d (float32 or float64 expected)", d.name, d.dtype);
            if (!ck.t.emplace(name, std::move(t)).second) fail("ev2h_pack_weights: duplicate key \"%s\"", name.c_str());
        }
        Folded F = fold_checkpoint(ck, in_channels);
        for (const auto& kv : ck.t)
            if (!kv.second.used) fail("ev2h_pack_weights: unexpected key \"%s\" in the checkpoint", kv.first.c_str());
        P = new ev2h_packed();
        memset(&P->w, 0, sizeof(P->w));
        P->w.precision = precision;
        if (precision == EV2H_PREC_F16) P->w.f16_families = ((flags >> 8) & 15) ? ((flags >> 8) & 15) : EV2H_FAM_ALL;
        // exact power-of-two channel equalisation: applied in EVERY precision mode, so that all modes run the same network
        // representation (the exact-fp32 results do not change by a bit)
        if (flags & EV2H_PACK_EQUALIZE) { P->eq = equalize_channels(F); P->w.flags |= EV2H_W_EQUALIZED; }
        if (flags & EV2H_PACK_UNEQUALIZED_OK) P->w.flags |= EV2H_W_UNEQUALIZED_OK;
        Builder b{*P, ns, {}};
        b.build(F, in_channels);
        const char* base = P->host.data();
        if (!(flags & EV2H_PACK_HOST_ONLY)) {
            P->dev_bytes = P->host.size();
            hipError_t e = hipMalloc(&P->dev, P->dev_bytes);
            if (e == hipSuccess) e = hipMemcpy(P->dev, P->host.data(), P->dev_bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                ev2h_set_error("ev2h_pack_weights: uploading %zu bytes failed: %s", P->host.size(), hipGetErrorString(e));
                if (P->dev) (void)hipFree(P->dev);
                delete P;
                return EV2H_ERR_HIP;
            }
            base = (const char*)P->dev;
        }
        for (auto& f : b.fix) *f.first = base + f.second;
        *out = P;
        return EV2H_OK;
    } catch (const std::exception& e) {
        delete P;
        ev2h_set_error("%s", e.what());
        return EV2H_ERR_ARG;
    }
}

extern "C" void ev2h_packed_free(ev2h_packed* p) {
    if (!p) return;
    if (p->dev) (void)hipFree(p->dev);
    delete p;
}

extern "C" const ev2h_weights* ev2h_packed_weights(const ev2h_packed* p) { return p ? &p->w : nullptr; }
extern "C" size_t ev2h_packed_bytes(const ev2h_packed* p) { return p ? p->host.size() : 0; }
extern "C" int ev2h_packed_tensor_count(const ev2h_packed* p) { return p ? (int)p->blobs.size() : 0; }

extern "C" int ev2h_packed_tensor(const ev2h_packed* p, int i, const char** name, int* rows, int* cols, int* dtype, const void** host,
                                  const void** device) {
    EV2H_CHECK_ARG(p && i >= 0 && i < (int)p->blobs.size());
    const Blob& b = p->blobs[i];
    if (name) *name = b.name.c_str();
    if (rows) *rows = b.rows;
    if (cols) *cols = b.cols;
    if (dtype) *dtype = b.dtype;
    if (host) *host = p->host.data() + b.off;
    if (device) *device = p->dev ? (const char*)p->dev + b.off : nullptr;
    return EV2H_OK;
}

extern "C" int ev2h_packed_equalization_count(const ev2h_packed* p) { return p ? (int)p->eq.size() : 0; }

extern "C" int ev2h_packed_equalization(const ev2h_packed* p, int i, const char** name, const double** e, int* n) {
    EV2H_CHECK_ARG(p && i >= 0 && i < (int)p->eq.size());
    if (name) *name = p->eq[i].first.c_str();
    if (e) *e = p->eq[i].second.data();
    if (n) *n = (int)p->eq[i].second.size();
    return EV2H_OK;
}

extern "C" int ev2h_packed_weight_spread_count(const ev2h_packed* p) { return p ? (int)p->wspread.size() : 0; }

extern "C" int ev2h_packed_weight_spread(const ev2h_packed* p, int i, const char** name, uint64_t counts[3]) {
    EV2H_CHECK_ARG(p && counts && i >= 0 && i < (int)p->wspread.size());
    const WSpread& w = p->wspread[i];
    if (name) *name = w.name.c_str();
    counts[0] = w.nonzero; counts[1] = w.low; counts[2] = w.high;
    return EV2H_OK;
}

extern "C" int ev2h_pack_sa_image_bytes(int C1, int C2, int C3, int planes, size_t out[2]) {
    EV2H_CHECK_ARG(out);
    int g[10];
    const int rc = ev2h_tile_geometry(C1, C2, C3, planes, g);
    if (rc) return rc;
    out[0] = (size_t)(C1 / 32) * g[4];
    out[1] = (size_t)(C3 / 32) * g[5];
    return EV2H_OK;
}

extern "C" int ev2h_pack_sa_images(const double* W2, const double* W3, int C1, int C2, int C3, int planes, void* img2, void* img3, float* u2,
                                   float* u3) {
    EV2H_CHECK_ARG(W2 && W3 && img2 && img3 && u2 && u3 && planes >= 1 && planes <= 4);
    EV2H_TRY
    const SaImages im = sa_images(W2, W3, C1, C2, C3, planes);
    memcpy(img2, im.i2.data(), im.i2.size());
    memcpy(img3, im.i3.data(), im.i3.size());
    *u2 = im.u2; *u3 = im.u3;
    return EV2H_OK;
    EV2H_CATCH
}

extern "C" size_t ev2h_pack_gemm_image_bytes(int N, int K, int planes, int tile_rows) {
    if (N <= 0 || K <= 0 || planes < 1 || planes > 4 || tile_rows <= 0) return 0;
    return (size_t)(up(N, tile_rows) / tile_rows) * (up(K, 32) / 32) * tile_rows * (npl(planes) * 64 + 16);
}

extern "C" int ev2h_pack_gemm_image(const double* W, int N, int K, int planes, int tile_rows, void* img, float* u) {
    EV2H_CHECK_ARG(W && img && u && N > 0 && K > 0 && planes >= 1 && planes <= 4 && (tile_rows == 128 || tile_rows == 256));
    EV2H_TRY
    const std::vector<uint8_t> im = gemm_image(W, N, K, planes, tile_rows, u);
    memcpy(img, im.data(), im.size());
    return EV2H_OK;
    EV2H_CATCH
}

extern "C" float ev2h_plane_unscale(const double* W, size_t count, int planes) {
    return (W && count) ? (float)plane_unscale(W, count, planes) : 1.f;
}
