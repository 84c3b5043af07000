// Fused grouped set-abstraction MLP on the 16-bit matrix pipe (v_mfma_f32_32x32x16_{bf16,f16}, 16x the fp32 MFMA
// rate), same structure and same reference lines as sa_mlp.hip (pointnet2_utils.py:244-257); operand planes: planes.hpp
//   NS = 2  "f16x2": two fp16 planes per operand, 3 plane products -- fp32-class accuracy at 3/16 of the fp32 MFMA cost
//   NS = 3  "bf16x3": every fp32 operand is split exactly into three bf16 planes (8+8+8 mantissa bits,
//           truncation split x = h + m + l) and the six products hh, hm, mh, mm, hl, lh are accumulated in
//           fp32 -- dropped terms are O(2^-24), i.e. fp32-class accuracy at 6/16 of the fp32 MFMA cost;
//   NS = 1  plain bf16 operands (round to nearest even), fp32 accumulate -- BASELINE.json config 3.
//   NS = 4  "f16" [r6]: ONE fp16 plane per operand (11 mantissa bits, one MFMA per multiply-add like bf16) with f16x2's range
//           machinery (range records, per-window powers of two, W / u planes); layer 1 is BF16's matrix-pipe form (L1M: ONE
//           MFMA per chunk, inputs as two fp16 planes, weights as one) under wave-uniform powers of two.  NS = 4 is a mode code:
//           plane_count(NS) = 1 planes are stored, planes_f16(NS) selects the fp16 MFMA and the range scaling (planes.hpp).
//           The set abstractions keep ONE power of two per window for the whole chain in this mode (C2ONE in the kernel): layer 2's
//           accumulators are converted as they are, as in BF16.
// All biases, ReLUs and the max stay in fp32.  Layer 1 has three forms (see L1M / L1F / the VALU path in the kernel):
//   * gathered layer-1 table row + exact fp32 relative-xyz fma chain (every mode when the features are a real table: enc.sa2);
//   * BF16: one MFMA per 32-channel chunk (inputs as two bf16 planes), on top of the table row or -- raw feature rows -- instead of it;
//   * F16X2 with raw feature rows: two MFMAs per chunk with per-neighbour power-of-two factors for the feature and xyz groups;
//   * BF16X3 with raw feature rows [r5]: three MFMAs per chunk = the six plane products, no factors (bf16 keeps the float's exponent).
// Other round-4 additions: fragment reads issued a group ahead (FRAG_PIPE), the next strip's gather requested during the current
// strip (XPF), a group's strips spread over the waves of a workgroup when the grid is smaller than the chip (SaBP::spg).
//
// Weight tiles are packed by the host as byte images of the LDS tiles (rows padded by 16 B so that the
// ds_read_b128 fragment reads are conflict free) and streamed global -> LDS with the LDS-DMA
// (global_load_lds_dwordx4, no staging registers): two buffers and one barrier per tile step (one or two tiles) in the
// streamed variant; the narrow MLPs keep all tiles resident in a persistent workgroup (see the kernel's comment).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "planes.hpp"
#include "ev2hands_hip.h"

namespace {

struct SaBP {
    const float* P1; int ldp;
    const float4* pts4;
    const float4* ctr4;
    const int32_t* gidx;
    const float4* W1x;
    const char* W2s; const float* b2;     // bf16 tile images
    const char* W3s; const float* b3;
    float* out; int ldo;
    int B, Npts, S, K;
    // a launch may cover the centroids [s_off, s_off + S) of every window of arrays laid out for S_total centroids per window
    // (ev2h_sa_desc.S_total / s_off [r6]: chunks of enc.sa1 run beside the rest of its farthest-point sampling); S_total == S: all
    int S_total, s_off;
    int nblk;
    const int32_t* cnt; int cnt_ld;       // optional distinct-neighbour counts (ev2h_sa_desc.cnt)
    float u2, u3;                         // power-of-two unscale factors of the W2s / W3s planes (ev2h_sa_desc.w2_unscale)
    int per_xcd;                          // resident variant: groups per XCD (multiple of 8); nblk is a multiple of 8
    // f16x2 activation range (ev2h_sa_desc; NULL p1_scale = off)
    const float* p1_scale; const unsigned* p1_amax;   // per window: power of two the P1 table was stored with, max |stored P1|
    float w1x_norm, dmax, w2_norm, b2_max;
    unsigned* out_amax;                    // per window: atomicMax of |out|
    // ROWS (feature propagation, ev2h_fp_mlp): a "group" is a strip of 32 consecutive points of a window; P1 is the coarse points'
    // layer-1 table, the row of a point is the 3-NN blend of three table rows; out holds one row per point
    const int32_t* nn_idx; const float* nn_w; int N;
    // MODE 2 (plain row chain, e.g. the segmentation head): P1 holds the N input rows themselves (no blend, no layer-1 ReLU; F16X2:
    // scaled here by the power of two of p1_amax); ncols = leading columns of the last tile that are written; relu_out = 0 drops
    // the last ReLU; out_cm = optional second copy of the output, channel-major [B][ncols][N]
    int ncols; int relu_out; float* out_cm; size_t out_cm_stride;
    // BF16 (NS = 1), set abstraction: layer 1 runs on the matrix pipe (see L1M in the kernel).  feat != NULL: the layer-1 table is
    // not needed at all -- the neighbour's raw feature row feat[b][idx][0..nfeat) (nfeat <= 5, row stride ldf floats) and its
    // relative xyz are contracted with [W1f | W1x | b1] directly (W1f [C1][ldw1f], b1 [C1]); feat == NULL: P1 is the table.
    const float* feat; int ldf; const float* W1f; int ldw1f; const float* b1; int nfeat;
    // F16X2 with raw feature rows: range record of the feature rows, the layer-1 bounds and the power-of-two plane factor of
    // [W1f | W1x] (ev2h_sa_desc)
    const unsigned* feat_amax; float w1f_norm, b1_max, u1f, u1x;
    // F16 (NS = 4), layer 1 on the matrix pipe (L1M): launch-constant powers of two of the three k-slot groups of the A tile --
    // A = [W1f / a1f | W1x / a1x | W1x / a1x | b1 / a1b | W1f / a1f], B = s1 [a1f f | a1x lo(d) | a1x hi(d) | a1b | a1f lo(f)] with the
    // window's power of two s1 (chosen from the layer's rigorous bound, so that every B slot stays below 2^8 and every A entry below
    // 2^9: see ev2h_sa_mlp_max_bf16)
    float a1f, a1x, a1b;
    // streamed set abstraction, small grids: spg > 1 = the K / 32 strips of a group are spread over spg waves of one workgroup (a
    // workgroup then holds 8 / spg groups) and their partial maxima are combined through LDS -- a max is exact and order-free, so
    // the result is bit-identical; the weights are streamed once per strip SET instead of once per strip of the longest group.
    // One window at a time (demo.py:24-33) has 16 workgroups for 256 CUs in the 128-centroid modules: 88 -> ~25 us per launch.
    int spg;
    float* xyz_out; int xyz_ld;            // optional: the group's centroid as 8 more columns of the consumer's input rows (ev2h_sa_desc)
    // BF16 row chains only (internal, ev2h_fp_mlp_ex): the input rows of MODE 2 / the output rows are bf16 (2 bytes per value; ldp /
    // ldo still count VALUES) -- l0, the one N-row 256-wide tensor of the forward, is written once and read three times, and
    // every reader of the BF16 mode rounds it to bf16 before it multiplies anyway
    int t_bf16, out_bf16;
    // F16 row chains only (internal, ev2h_fp_mlp_ex) [r6]: the same for the one-plane fp16 mode -- the rows are fp16 values TIMES a
    // per-window power of two row16_scale[b] (float [B]).  The writer (out_bf16, fp1) chooses it from the rigorous bound of its own
    // output (|W3|_1 bound(H2) + max|b3|: known before the first row is computed), stores it and records max |stored|; the reader
    // (t_bf16, the segmentation head) takes the stored values as its operand plane as they are and the scale as its s1.
    float* row16_scale; float w3_norm, b3_max;
};

// EV2H_SAB_TIMELINE build (EV2H_BUILD_DEFS=-DEV2H_SAB_TIMELINE python -m ev2hands_amd.build --force; tools/sa_timeline.py):
// wave 0 of one workgroup records s_memtime at the phase boundaries of its second strip (costs ~10 % of the kernel time)
#ifdef EV2H_SAB_TIMELINE
__device__ long long g_sab_timeline[256];
#define STAMP(i) do { if (dbgw && strip == dbg_strip) g_sab_timeline[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
// The table form and the feature-row form of the set abstraction (ev2h_sa_desc.feat) are SEPARATE INSTANTIATIONS (MODE 0 / MODE 3) [r5]:
// one body that chooses at run time keeps both forms' registers live (128-196-256: 249 registers instead of 224 / 214 in F16X2, a
// 92-byte spill in BF16X3; the BF16 feature-row kernels 158 -> 124, which doubles the resident workgroups per CU) and schedules around
// the branch.  Same-box: F16X2 step +2.3 %, BF16 +4 % (profiles/r5_ab_split_forms.txt).
// EV2H_BUILD_DEFS=-DEV2H_NO_SPLIT_FORMS: BF16 and F16X2 as one body again (A/B; BF16X3 has no such build -- it would spill).
#ifdef EV2H_NO_SPLIT_FORMS
constexpr bool SAB_SPLIT12 = false;
#else
constexpr bool SAB_SPLIT12 = true;
#endif
template <int NS> constexpr bool sab_split_forms() { return NS == 3 || SAB_SPLIT12; }
constexpr int SAB_WAVES = 8;
constexpr int SAB_THREADS = SAB_WAVES * 64;

template <int C1, int C2, int C3, int NS>
struct SaBCfg {
    static constexpr int NPL = plane_count(NS);                         // 16-bit planes per operand in the tile images
    static constexpr bool F16 = planes_f16(NS);                         // fp16 planes: range records and power-of-two scaling
    static constexpr int T2 = (C2 + 31) / 32;
    static constexpr int REM = C2 % 32;
    static constexpr int M_LAST = REM == 0 ? 2 : (REM <= 16 ? 1 : 2);   // live 16-wide k-blocks of the last layer-2 tile
    static constexpr int C2P = 32 * (T2 - 1) + 16 * M_LAST;              // layer-3 contraction length (permuted order)
    static constexpr int T3 = C3 / 32;
    static constexpr int NC1 = C1 / 32;
    static constexpr int RS2 = NPL * 64 + 16;                             // bytes per row of a W2 chunk tile
    static constexpr int RS3 = NPL * C2P * 2 + 16;                        // bytes per row of a W3 tile
    static constexpr int TB2 = T2 * 32 * RS2;
    static constexpr int TB3 = 32 * RS3;
    // streamed variant: a tile step moves CPT layer-2 chunk tiles or UPT layer-3 tiles at once (one DMA burst, one barrier);
    // two per step whenever the doubled buffers still fit in LDS (measured on 128-128-256 f16x2: -16 % with 2, -11 % with 4)
    static constexpr int SB2W = F16 ? SAB_WAVES * T2 * 32 * 4 : 0;   // F16X2: the b2 bias times each wave's window scale
    // W1x in fp32 (12 B per channel, padded); BF16: the layer-1 A tile [C1][16 k] in bf16; F16X2: two A tiles per channel row
    // ([wh | wh], [wl | 0]: 64 B) + the b1 bias times each wave's window scale
    // BF16X3: three A tiles per channel row ([w0 | w0], [w1 | w1], [w2 | w0]: 96 B) + the b1 bias
    // [r6] the L1F A rows carry a 16-byte pad (80 / 112 instead of 64 / 96 bytes): at 64 bytes per row the 32 rows a fragment read
    // touches fall on 4 of the LDS's 256-byte bank windows eight deep -- SQ_LDS_BANK_CONFLICT 15.6 M cycles per launch in the
    // feature-row form against 0 in the table form (profiles/r5_pmc_sq_f16x2.txt); padded like the weight tiles they are conflict free.
    // EV2H_BUILD_DEFS=-DEV2H_L1F_NO_PAD: the unpadded rows (A/B).
#ifdef EV2H_L1F_NO_PAD
    static constexpr int RSA = (NS == 3) ? 96 : 64;
#else
    static constexpr int RSA = (NS == 3) ? 112 : 80;
#endif
    static constexpr int W1B = (NS == 1 || NS == 4) ? C1 * 32 : F16 ? C1 * RSA + SAB_WAVES * C1 * 4 : C1 * RSA + C1 * 4;
    // range-record combine (F16X2, end of a group): 8 x (window, max) in the streamed variants, REC_SLOTS per-window running maxima
    // in the resident one -- see the kernel's epilogue
    static constexpr int REC_SLOTS = 64;
    static constexpr int REC = F16 ? REC_SLOTS * 4 : 0;
    static constexpr int SMALL_NOREC = W1B + T2 * 32 * 4 + 16 + SB2W;      // + b2, one int for the workgroup's strip count, scaled b2
    static constexpr int SMALL = SMALL_NOREC + REC;
    static constexpr int tile_bytes(int cpt, int upt) { return ((cpt * TB2 > upt * TB3 ? cpt * TB2 : upt * TB3) + 1023) / 1024 * 1024; }
    static constexpr bool fits(int n) { return NC1 % n == 0 && T3 % n == 0 && 2 * tile_bytes(n, n) + SMALL <= 158 * 1024; }
    static constexpr int CPT = fits(2) ? 2 : 1, UPT = CPT;
    static constexpr int TILE = tile_bytes(CPT, UPT);
    static constexpr int LDS_BYTES = 2 * TILE + SMALL;
    // resident variant: every tile image of the module stays in LDS for the lifetime of the (persistent) workgroup
    static constexpr int RES_W = (NC1 * TB2 + T3 * TB3 + 1023) / 1024 * 1024;
    static constexpr int RES_LDS_BYTES = RES_W + SMALL;
    static constexpr bool FITS_RESIDENT = RES_LDS_BYTES <= 160 * 1024 && C1 >= 64;   // (the 32-32-64 MLP is faster streamed: several small workgroups per CU)
    // F16X2 with 1..4 channels left over after the last full 32-channel tile (C2 = 196: channels 192..195): the three plane
    // products of those channels share MFMAs instead of taking three each (csrc/pack.hip builds the matching images):
    //  * layer 2, last tile: rows 8..11 of the tile's HIGH-plane image hold the LOW plane of rows 0..3, so (A = high image,
    //    B = xh) yields wh*xh in D rows 0..3 and wl*xh in rows 8..11 -- the separate (wl, xh) MFMA is dropped and rows 8..11 are
    //    added to rows 0..3 in registers (same lane: D rows 8..11 are registers 4..7 of the lower half-wave);
    //  * layer 3, last 16-slot k-block (4 live slots): the slots carry [xh | xl | xh | 0] against [wh | wh | wl | 0] -- ONE MFMA
    //    instead of three.
    // 24 of the 480 MFMAs of a 128-196-256 strip (5 %) disappear; the products are the same three (plus wl*xl in layer 2).
#ifdef EV2H_NO_PACK4
    static constexpr bool PACK4 = false;
#else
    static constexpr bool PACK4 = (NS == 2) && REM >= 1 && REM <= 4;
#endif
};

// RES = false: weight tiles are streamed, two LDS buffers, one barrier per tile (any MLP width).
// RES = true : all tile images of the module fit in LDS (the three narrow MLPs of sa1): they are loaded once by a
//              persistent workgroup whose waves then walk their groups with NO barrier and no DMA in the loop -- the
//              waves drift apart, so one wave's split (VALU) phases run under the other waves' MFMA phases, and a tile
//              step no longer pays the DMA issue / wait / barrier that dominate when a tile holds only 6-18 MFMAs.
// MODE = 1, 2 (streamed only): the row chains of ev2h_fp_mlp -- same tile walk, but the layer-3 tiles are written out row by row
//              instead of maximised, and layer 1 is (1) the inverse-distance blend of three table rows (no relative xyz) or (2) the
//              input row itself.
// Waves per SIMD the register allocator must leave room for: 2 (256 registers) everywhere, except F16's resident feature-row kernels
// whose LDS footprint lets TWO workgroups share a CU -- BF16's get under 128 registers by themselves (124), F16's 64-96-128 needs 138
// and is held to 128 (32 bytes of scratch outside the MFMA loops): kbench 0.81 -> 0.73 ms (tools/kbench.py sab, KBENCH_FEAT=1).
// EV2H_BUILD_DEFS=-DEV2H_F16_RES_ONE_WG: build without (A/B).
template <int C1, int C2, int C3, int NS, bool RES, int MODE>
constexpr int sab_min_waves() {
#ifdef EV2H_F16_RES_ONE_WG
    return 2;
#else
    return (NS == 4 && RES && MODE == 3 && 2 * SaBCfg<C1, C2, C3, NS>::RES_LDS_BYTES <= 160 * 1024) ? 4 : 2;
#endif
}
template <int C1, int C2, int C3, int NS, bool RES, int MODE = 0>
__global__ __launch_bounds__(SAB_THREADS, (sab_min_waves<C1, C2, C3, NS, RES, MODE>())) void sa_mlp_max_bf16_kernel(SaBP p) {
    constexpr int WV = SAB_WAVES;
    // MODE = 3: the set abstraction with RAW FEATURE ROWS compiled in (L1M / L1F below; MODE 0 is then the table form only) -- see
    //              sab_split_forms above.
    constexpr bool ROWS = MODE == 1 || MODE == 2, DIRECT = MODE == 2;
    static_assert(MODE != 3 || sab_split_forms<NS>(), "MODE 3 is the feature-row form of the modes that compile it separately");
    static_assert(!(RES && ROWS), "the row-output variants stream their tiles");
    // BF16, set abstraction: LAYER 1 ON THE MATRIX PIPE.  D1[channel][neighbour] = A1 [32 channels][16 k] x B1 [16 k][32 neighbours]
    // (+ C = the gathered table row when the features are a table) with the k slots
    //     0..4  feature f_i (hi plane)      5..7  lo(dx, dy, dz)      8..10  hi(dx, dy, dz)      11  1.0 (x b1)      12..15  lo(f_0..f_3)
    // (hi = bf16(x), lo = bf16(x - hi): inputs keep 16 bits, the weights are bf16 like everywhere in this mode) -- one MFMA per
    // 32-channel chunk instead of 3 fma + ReLU + convert per (channel, neighbour) pair (288 -> ~100 VALU instructions per strip),
    // and with raw feature rows (<= 5 channels: enc.sa1, the regressors' sa1) no layer-1 table exists at all.  The D registers
    // of a lane are channels 8q + 4 half + e, so the layer-2 k slot (block m, half h, e) is channel 16m + 4h + (e & 3) + 8(e >> 2):
    // the BF16 W2 images are stored in that order (ev2h_tile_geometry out[9] = 1), for every kernel mode.
    // F16 (NS = 4) [r6]: the same form in fp16 -- the k-slot groups carry launch-constant powers of two (SaBP::a1f / a1x / a1b), the B
    // operand the window's s1, so that D1 = s1 (W1f f + W1x d + b1) comes out of the MFMA ready for ReLU + conversion (2 VALU ops per pair).
    constexpr bool L1M = (NS == 1 || NS == 4) && !ROWS;
    // F16X2, set abstraction with RAW FEATURE ROWS (ev2h_sa_desc.feat): layer 1 on the matrix pipe with a power-of-two scale PER
    // NEIGHBOUR.  The inputs of neighbour j -- features f0..f4 and relative xyz -- are two groups with their own weights (W1f / uf,
    // W1x / ux, planes built here) and their own input factors sigma_f = kappa_j uf, sigma_x = kappa_j ux, where
    // kappa_j = min(s_f / uf, s_x / ux) and s_g puts the group's largest input at [2^14, 2^15): both groups accumulate kappa_j times
    // their term in ONE accumulator, the group with the larger (input x weight) bound sits at full scale and the other keeps 22 bits
    // relative to that bound -- what fp32 itself does.  (One factor for all 8 inputs was not enough: a checkpoint with 1e6 x larger
    // hidden features and 1e-6 x smaller feature weights left the coordinates 2^-23 below the features and cost 2e-5, fuzz case
    // profiles/r4_fuzz_modes.txt.)  B1 = [xh(8) | xl(8)], A = [wh | wh] and [wl | 0]: two MFMAs per 32-channel chunk give the three plane
    // products, and since the neighbour is the N index of the product its column is un-scaled per LANE:
    // s1 H1 = relu(D1 (s1 / kappa_j) + s1 b1).  A neighbour's inputs are scaled by its OWN maxima -- a 1e7-event hot pixel no longer
    // sets the scale of the other neighbours, which is what the exact fp32 layer 1 used to guarantee -- no layer-1 table is
    // computed, stored or gathered (48 B instead of 512 B per neighbour), and one fma replaces three.
    // BF16X3 with raw feature rows [r5]: the same without any factor.  B1f = [x0(8) | x1(8)], B1g = [x0 | x2] (x = x0 + x1 + x2, the exact
    // three-plane split), A = [w2 | w0] (x B1g), [w1 | w1], [w0 | w0] (x B1f): three MFMAs = the six products of Planes<3>, small terms
    // first, on top of C = b1.
    constexpr bool L1F = !ROWS && (NS == 2 || NS == 3) && (!sab_split_forms<NS>() || MODE == 3);
    // L2PIPE / H2FUSE (round 4, late): conversion work of one wave placed between its MFMA groups (see the layer-2 loop and the first
    // layer-3 step).  Same-box step A/B (profiles/r4_ab_h2fuse_l2pipe.txt): BF16 +1.3 % with H2FUSE, +1.9 % with both; F16X2 +1.1 % with
    // H2FUSE, and L2PIPE costs it 0.5 % (its widest instantiation then spills 24 bytes) -- so F16X2 keeps the un-pipelined layer 2.
    // EV2H_BUILD_DEFS=-DEV2H_NO_L2PIPE / -DEV2H_NO_H2FUSE: build without.
#ifdef EV2H_NO_L2PIPE
    constexpr bool L2PIPE = false;
#else
#ifdef EV2H_L2PIPE2
    constexpr bool L2PIPE = !ROWS && NS <= 2 && (C1 / 32) % 2 == 0;      // build experiment: F16X2 too
#else
#ifdef EV2H_F16_NO_L2PIPE
    constexpr bool L2PIPE = !ROWS && NS == 1 && (C1 / 32) % 2 == 0;
#else
    constexpr bool L2PIPE = !ROWS && (NS == 1 || NS == 4) && (C1 / 32) % 2 == 0;      // (F16 [r6]: bf16's layer-1 form, bf16's pipelining)
#endif
#endif
#endif
#ifdef EV2H_NO_H2FUSE
    constexpr bool H2FUSE = false;
#else
    constexpr bool H2FUSE = NS != 3;          // (BF16X3: the widest instantiation would spill)
#endif
    // F16X2 [r5]: with the two layer-1 forms in separate instantiations there are registers for the second fragment set (128-196-256:
    // 224 -> 232): dominant launch 1.622 -> 1.592 ms, step +0.65 % same-box (profiles/r5_ab_frag_pipe_f16x2.txt).  BF16X3 would spill
    // (28 bytes in the widest instantiation).  EV2H_BUILD_DEFS=-DEV2H_NO_FRAG_PIPE2: F16X2 without (A/B).
#ifdef EV2H_NO_FRAG_PIPE2
    constexpr bool FRAG_PIPE = (NS == 1 || NS == 4);
#else
    constexpr bool FRAG_PIPE = (NS != 3);
#endif
    using Cfg = SaBCfg<C1, C2, C3, NS>;
    using PL = Planes<NS>;
    // L3T16 [r6]: LAYER 3 ON v_mfma_f32_16x16x32 (planes.hpp: mfma16_planes).  A pure stream of that instruction sustains 0.73-0.77 of the
    // 2.5 PFLOP/s peak on random operands where the 32 x 32 x 16 form sustains 0.62-0.65 (same FLOP per cycle, a quarter of the
    // accumulator traffic: the power limit sets the clock, profiles/r6_mfma_ceiling.txt), and layer 3 is 62 % of this kernel's MFMA work.
    // Layers 1-2 keep the 32 x 32 form and its D layout (lane = neighbour j, registers = channels); the A operand of a 16 x 16 x 32 MFMA
    // wants lane group G = l >> 4 to hold k-group G of neighbour l & 15, which is a regrouping of 16-lane rows between the two k-block
    // registers of a tile: P = h2p[.][t][0] = rows [P0 P1 P2 P3], Q = h2p[.][t][1]; v_permlane32_swap then v_permlane16_swap (gfx950,
    // one VALU op each; tools/ubench/permlane_swap_check.hip) give [P0 P2 Q0 Q2] = neighbours 0..15 x k-groups 0..3 and [P1 P3 Q1 Q3]
    // = neighbours 16..31.  The W3 images are read as they are (16-row fragments: row l & 15, 16 bytes at k offset 32 t + 8 G).
    // Same products, same plane order; the sums associate differently (32 k per MFMA, two neighbour tiles), so results agree with
    // the 32 x 32 form to fp32 rounding, not bit for bit (all 100 set-abstraction operator tests pass with it).
    // MEASURED AND NOT ADOPTED (profiles/r6_ab_l3t16_refuted.txt): the kernels are no faster (kbench: 1.653 / 1.807 / 1.747 ms against
    // 1.638 / 1.835 / 1.756 for the three 128-196-256 shapes) and the whole step is 1.6 % (f16x2) / 2.0 % (f16) SLOWER, same box,
    // builds interleaved -- the pure-stream advantage of the smaller tile does not survive the kernel's own mix (112 more VALU ops
    // per strip for the regrouping, twice the MFMA issue slots).  Opt-in for whoever re-tiles layers 1-2 as well (then the
    // regrouping disappears): EV2H_BUILD_DEFS=-DEV2H_L3T16.
#ifdef EV2H_L3T16
    constexpr bool L3T16 = !ROWS && (NS == 2 || NS == 4) && (RES || Cfg::UPT == 2);
#else
    constexpr bool L3T16 = false;
#endif
    constexpr int NPL = Cfg::NPL;
    constexpr bool F16 = Cfg::F16;
    constexpr int T2 = Cfg::T2, T3 = Cfg::T3, NC1 = Cfg::NC1, RS2 = Cfg::RS2, RS3 = Cfg::RS3, C2P = Cfg::C2P;
    constexpr int WBYTES = RES ? Cfg::RES_W : 2 * Cfg::TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wt0 = smem;
    char* wt1 = smem + Cfg::TILE;
    f32x4* sW1xT = reinterpret_cast<f32x4*>(smem + WBYTES);     // per 4 channels: x[4], y[4], z[4]
    float* sb2 = reinterpret_cast<float*>(smem + WBYTES + Cfg::W1B);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int L = xcd_remap(blockIdx.x, p.nblk);
    const int ngroups = p.B * p.S;
    // hasfeat: the features are raw rows (ev2h_sa_desc.feat) -- a compile-time constant where the two forms are separate instantiations
    const bool hasfeat = sab_split_forms<NS>() ? (MODE == 3) : (p.feat != nullptr);
    const bool fmode = L1F && hasfeat;                         // (uniform)
    // F16X2 feature mode: s1 b1 of this wave's window; BF16X3 feature mode: b1 (one copy)
    float* sb1w = F16 ? reinterpret_cast<float*>(smem + WBYTES + C1 * Cfg::RSA) + wave * C1 : reinterpret_cast<float*>(smem + WBYTES + C1 * Cfg::RSA);

    if constexpr (L1M) {
        // A1 [C1][16 k] bf16 (one 32-byte row per channel), k slots as listed above
        const float iaf = F16 ? 1.f / p.a1f : 1.f, iax = F16 ? 1.f / p.a1x : 1.f, iab = F16 ? 1.f / p.a1b : 1.f;      // (powers of two: exact)
        for (int i = tid; i < C1; i += WV * 64) {
            const float4 w = p.W1x[i];
            float k[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) k[j] = 0.f;
            k[5] = k[8] = w.x * iax; k[6] = k[9] = w.y * iax; k[7] = k[10] = w.z * iax;
            if (hasfeat) {
                for (int j = 0; j < p.nfeat; ++j) k[j] = p.W1f[(size_t)i * p.ldw1f + j] * iaf;
#pragma unroll
                for (int j = 0; j < 4; ++j) k[12 + j] = k[j];
                k[11] = p.b1[i] * iab;
            }
            unsigned* d = reinterpret_cast<unsigned*>(smem + WBYTES) + i * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { unsigned o[1]; split_planes<NS>(k[2 * j], k[2 * j + 1], o); d[j] = o[0]; }
        }
    } else if constexpr (!ROWS) {
        if (fmode) {
            // A tiles of layer 1: row i = [wh(v0..v7) | wh(v0..v7)] then [wl(v0..v7) | 0], v = (f0..f4, dx, dy, dz), planes of W1f / uf, W1x / ux
            const float iu = F16 ? 1.f / p.u1f : 1.f, iux = F16 ? 1.f / p.u1x : 1.f;      // (powers of two: exact; BF16X3: no plane factor)
            for (int i = tid; i < C1; i += WV * 64) {
                const float4 w = p.W1x[i];
                float k[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) k[j] = 0.f;
                for (int j = 0; j < p.nfeat; ++j) k[j] = p.W1f[(size_t)i * p.ldw1f + j] * iu;
                k[5] = w.x * iux; k[6] = w.y * iux; k[7] = w.z * iux;
                if constexpr (F16) {          // (F16 too: layer 1 keeps the two-plane form)
                    unsigned* d = reinterpret_cast<unsigned*>(smem + WBYTES + i * Cfg::RSA);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned o[2];
                        split_planes<2>(k[2 * j], k[2 * j + 1], o);
                        d[j] = o[0]; d[4 + j] = o[0]; d[8 + j] = o[1]; d[12 + j] = 0u;
                    }
                } else if constexpr (NS == 3) {
                    unsigned* d = reinterpret_cast<unsigned*>(smem + WBYTES + i * Cfg::RSA);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned o[3];
                        split_planes<3>(k[2 * j], k[2 * j + 1], o);
                        d[j] = o[0]; d[4 + j] = o[0]; d[8 + j] = o[1]; d[12 + j] = o[1]; d[16 + j] = o[2]; d[20 + j] = o[0];
                    }
                    sb1w[i] = p.b1[i];
                }
            }
        } else {
            for (int i = tid; i < C1; i += WV * 64) {
                const float4 w = p.W1x[i];
                float* d = reinterpret_cast<float*>(sW1xT) + (i >> 2) * 12 + (i & 3);
                d[0] = w.x; d[4] = w.y; d[8] = w.z;
            }
        }
    }
    for (int i = tid; i < T2 * 32; i += WV * 64) sb2[i] = p.b2[i] / p.u2;     // accumulators hold (W2 h1 + b2) / u2 (exact: power of two)
    // F16X2: the accumulators hold (s1 / u2)(W2 h1 + b2) with the window's power of two s1; each wave keeps b2 s1 / u2 of its
    // window here, so that an accumulator tile is initialised by four LDS reads and no arithmetic
    float* sbw = F16 ? reinterpret_cast<float*>(smem + WBYTES + Cfg::W1B + T2 * 32 * 4 + 16) + wave * (T2 * 32) : sb2;

    // LDS-DMA of one tile image: each wave instruction moves 1 KiB (64 lanes x 16 B), lane-linear on both sides
    auto dma_tile = [&](const char* src, char* dst, int bytes) {
        for (int off = wave * 1024; off < bytes; off += WV * 1024) {
            if (off + lane * 16 < bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + off), 16, 0, 0);
        }
    };
    constexpr int CPT = RES ? 1 : Cfg::CPT, UPT = RES ? 1 : Cfg::UPT;
    auto dma_w2 = [&](int step, char* dst) { dma_tile(p.W2s + (size_t)step * CPT * Cfg::TB2, dst, CPT * Cfg::TB2); };
    auto dma_w3 = [&](int step, char* dst) { dma_tile(p.W3s + (size_t)step * UPT * Cfg::TB3, dst, UPT * Cfg::TB3); };

    int buf = 0;
    if constexpr (RES && F16) {
        if (tid < Cfg::REC_SLOTS) reinterpret_cast<unsigned*>(smem + WBYTES + Cfg::SMALL_NOREC)[tid] = 0u;      // per-window running maxima of the record combine
    }
    if constexpr (RES) {
        dma_tile(p.W2s, smem, NC1 * Cfg::TB2);
        dma_tile(p.W3s, smem + NC1 * Cfg::TB2, T3 * Cfg::TB3);
    } else {
        dma_w2(0, wt0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // RES: XCD x (= blockIdx & 7, the dispatcher's round-robin) owns the contiguous groups [x * per_xcd, (x + 1) * per_xcd);
    // its nblk/8 workgroups sweep that range together, 8 groups per workgroup per step, so that at any time one XCD works on
    // a few neighbouring windows whose P1 tables and index lists stay in its L2 (a workgroup-contiguous split measured 4.5x
    // the algorithmic HBM traffic: 32 windows in flight per XCD do not fit the 4 MB L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nbx = p.nblk >> 3;
    const int g_end = RES ? min(ngroups, (xcd + 1) * p.per_xcd) : ngroups;
  const int spg = (RES || ROWS) ? 1 : max(1, __builtin_amdgcn_readfirstlane(p.spg)), sw = wave % spg;      // sw: this wave's strip of its group
  for (int g = (RES ? xcd * p.per_xcd + slot * WV : (spg > 1 ? L * (WV / spg) + wave / spg : L * WV)) + wave * (RES || spg == 1); RES ? g < g_end : true;
       g += nbx * WV) {
    const bool valid = g < ngroups;
    const int gg = valid ? g : ngroups - 1;
    const int b = gg / p.S;
    // index of this group in the caller's arrays (centroids, index lists, counts, output rows)
    const size_t gf = ROWS ? (size_t)gg : (size_t)b * p.S_total + p.s_off + (gg - b * p.S);
    float mrun[T3];
    float m16[L3T16 ? T3 : 1][2];          // L3T16: running maxima per 16-channel half of an output tile (this lane's column l & 15)
#pragma unroll
    for (int u = 0; u < (L3T16 ? T3 : 1); ++u) m16[u][0] = m16[u][1] = -INFINITY;
#pragma unroll
    for (int u = 0; u < T3; ++u) mrun[u] = -INFINITY;

    float4 ctr = make_float4(0.f, 0.f, 0.f, 0.f);
    const int32_t* gi = nullptr;
    if constexpr (!ROWS) { ctr = p.ctr4[gf]; gi = p.gidx + gf * p.K + sw * 32; }
    const int row0 = ROWS ? (gg - b * p.S) * 32 : 0;          // ROWS: first point of this strip inside its window
    unsigned am = 0u;
    // f16x2 range: with P1' = s1 P1 (table stored scaled) and d' = s1 d the layer-1 output is H1' = s1 H1 <= a1 + s1 |W1x|_1 dmax
    // (< 2^15 by the table's choice of s1); layer 2 accumulates (s1 / u2)(W2 H1 + b2); H2' = s2 H2 with the power of two s2 that
    // keeps the bound |W2|_1 max(H1) + max|b2| below 2^15; layer 3 accumulates (s2 / u3) W3 H2.  All factors are exact.
    // [r6] F16 set abstractions (one fp16 plane): ONE power of two per window for the whole chain.  s1 is chosen (here for raw feature rows,
    // by the table's producer otherwise) so that H1' = s1 H1 AND H2' = (s1 / u2) H2 -- layer 2's accumulators as they are -- stay below
    // 2^15; with u2 = 2^floor(log2 |W2|_1) (pack.hip: chain_unscale) the two bounds agree to a factor 2-4, and one plane loses nothing to a
    // few binades of headroom.  The conversion of H2 is then cvt + packed ReLU + packed clamp, without the two multiplies by c2
    // (whole step +6.7 % in a timing build, profiles/r6_f16_chain_scale.txt).  The row chains (ROWS) read rows whose scale
    // their producer chose for the rows alone: they keep the factor.
#ifdef EV2H_F16_CLAMP
    constexpr bool F16_CLAMP = true;              // EV2H_BUILD_DEFS=-DEV2H_F16_CLAMP: clamp the fp16 conversions of the F16 mode at 65504 (A/B)
#else
    constexpr bool F16_CLAMP = false;
#endif
#ifdef EV2H_F16_NO_C2ONE
    constexpr bool C2ONE = false;                 // EV2H_BUILD_DEFS=-DEV2H_F16_NO_C2ONE: a factor per layer as in f16x2 (A/B)
#else
    constexpr bool C2ONE = NS == 4 && !ROWS;
#endif
    float s1 = 1.f, c2 = p.u2, c3 = C2ONE ? p.u3 * p.u2 : p.u3;      // (without range arguments: s1 = 1; C2ONE: H2' = H2 / u2 as accumulated)
    float so = 1.f;                                             // F16 row chain with fp16 output rows: their per-window power of two
    float b1s_f = 1.f, b1s_x = 1.f, b1s_b = 1.f;               // F16 L1M: the B operand's factors s1 a1f, s1 a1x, s1 a1b (wave-uniform)
    const bool fform = fmode || (L1M && hasfeat);               // layer 1 reads raw feature rows (no table, no producer that chose s1)
    if constexpr (F16) {
        if (fform && p.feat_amax) {
            // no table and no producer that chose s1: the same bound, evaluated here from the record of the feature rows
            const float bnd = __fmaf_rn(p.w1f_norm, __uint_as_float(p.feat_amax[b]), p.b1_max) + p.w1x_norm * p.dmax;
            s1 = f16x2_scale(__float_as_uint(bnd));
            float s2 = f16x2_scale(__float_as_uint(__fmaf_rn(p.w2_norm, bnd, p.b2_max)));
            if constexpr (C2ONE) { s1 = fminf(s1, p.u2 * s2); s2 = s1 * pow2_inverse(p.u2); }     // one power of two for the chain: c2 = 1
            c2 = p.u2 * s2 * pow2_inverse(s1);
            c3 = p.u3 * pow2_inverse(s2);
        } else if (!fform && p.p1_amax) {
            float a1 = __uint_as_float(p.p1_amax[b]);
            if constexpr (DIRECT) {
                if (NS == 4 && p.t_bf16) s1 = p.row16_scale[b];                        // fp16 rows: stored with this power of two, the record is of the stored values
                else { s1 = f16x2_scale(p.p1_amax[b]); a1 *= s1; }                     // unscaled input rows: scaled as they are read
            }
            else s1 = p.p1_scale[b];
            const float inv_s1 = pow2_inverse(s1);
            const float bh1 = a1 + s1 * (p.w1x_norm * p.dmax);
            const float bh2 = __fmaf_rn(p.w2_norm, bh1 * inv_s1, p.b2_max);
            float s2 = f16x2_scale(__float_as_uint(bh2));
            c2 = p.u2 * s2 * inv_s1;
            c3 = p.u3 * pow2_inverse(s2);
            if constexpr (C2ONE) {
                // the table's producer chose s1 for the whole chain (ev2h_sa_desc.p1_scale, F16 contract): H2' = (s1 / u2) H2 as accumulated.
                // A scale that breaks the contract would saturate hidden values silently: the window's output is NaN instead.
                s2 = s1 * pow2_inverse(p.u2);
                c3 = (s2 * bh2 <= 65504.f) ? p.u3 * pow2_inverse(s2) : __uint_as_float(0x7fc00000u);
            }
        }
        // wave-uniform by construction (one group per wave): keep the three factors in scalar registers
        s1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s1)));
        c2 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(c2)));
        c3 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(c3)));
        if constexpr (ROWS && NS == 4) {
            if (p.out_bf16) {
                // fp16 output rows: out' = so out with the power of two so that keeps |W3|_1 bound(H2) + max|b3| below 2^15, bound(H2) =
                // 2^15 / s2 by the choice of s2 (c3 = u3 / s2).  Folded into the epilogue's factor and bias: out' = acc (c3 so) + b3 so.
                const float bound2 = 32768.f * (c3 / p.u3);                             // = 2^15 / s2 >= bound(H2)
                so = f16x2_scale(__float_as_uint(__fmaf_rn(p.w3_norm, bound2, p.b3_max)));
                so = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(so)));
                if (lane == 0 && valid && row0 == 0) p.row16_scale[b] = so;            // (one writer per window: its first strip)
            }
        }
        for (int i = lane; i < T2 * 32; i += 64) sbw[i] = sb2[i] * s1;           // read back by this wave only (LDS ops of a wave stay in order)
        if constexpr (L1F) {
            if (fmode) for (int i = lane; i < C1; i += 64) sb1w[i] = p.b1[i] * s1;
        }
        if constexpr (L1M) { b1s_f = s1 * p.a1f; b1s_x = s1 * p.a1x; b1s_b = s1 * p.a1b; }      // (SGPRs: s1 is, the a1 are kernel arguments)
    }
    // Slots >= cnt of a group repeat slot 0 (ball-query padding, pointnet2_utils.py:104-106): 32-slot strips made only of
    // padding cannot change the max and are skipped.  In the streamed variant every wave still walks the tile steps of the
    // workgroup's longest group (DMA pieces and barriers), without computing.
    int my_strips = ROWS ? 1 : p.K >> 5;
    if (!ROWS && p.cnt) my_strips = min(my_strips, max(1, (p.cnt[gf * p.cnt_ld] + 31) >> 5));
    if (spg > 1) my_strips = (valid && sw < my_strips) ? 1 : 0;      // spread: this wave owns strip sw of its group (or nothing)
    int nstrips = my_strips;
    if constexpr (!RES && !ROWS) {
        int* s_strips = reinterpret_cast<int*>(smem + WBYTES + Cfg::W1B + T2 * 32 * 4);
        if (tid == 0) *s_strips = 1;
        __syncthreads();
        if (lane == 0) atomicMax(s_strips, my_strips);
        __syncthreads();
        nstrips = *s_strips;
    }

#ifdef EV2H_SAB_TIMELINE
    const bool dbgw = (blockIdx.x == 300 && tid == 0);
    const int dbg_strip = nstrips > 1 ? 1 : 0;
#endif
    // XPF (set abstraction): the neighbour gather of strip s + 1 -- index, then coordinates and feature row (or the first
    // table chunk) -- is requested DURING strip s, so that a strip no longer starts with two dependent global-memory latencies
    // (index -> row: 6 of the 28 us of a strip in the phase timeline, with one workgroup per CU and nothing else to run)
    // (F16X2 and BF16: +0.9 % on the f16x2 step, dominant kernel 1.620 -> 1.576 ms, same-box build A/B profiles/r4_ab_xpf.txt;
    // BF16X3 would spill: its widest instantiation already sits at 256 registers.)  EV2H_BUILD_DEFS=-DEV2H_NO_XPF: build without.
#ifdef EV2H_NO_XPF
    constexpr bool XPF = false;
#else
    constexpr bool XPF = !ROWS && NS != 3;
#endif
    f32x4 raw[4];                 // a lane's 16 gathered layer-1 values of the current chunk (XPF: survives into the next strip)
    int idx_cur = 0, idx_nxt = 0;
    float4 q_cur = make_float4(0.f, 0.f, 0.f, 0.f), f0_cur = q_cur, f1_cur = q_cur;
    if constexpr (XPF) {
        idx_cur = gi[l31];
        q_cur = p.pts4[(size_t)b * p.Npts + idx_cur];
        if (hasfeat) {
            const float4* fr = reinterpret_cast<const float4*>(p.feat + ((size_t)b * p.Npts + idx_cur) * p.ldf);
            f0_cur = fr[0]; f1_cur = fr[1];
        }
    }
    for (int strip = 0; strip < nstrips; ++strip) {
        STAMP(0);
        if constexpr (!RES) {
            if (strip >= my_strips) {           // wave-uniform: keep the workgroup's DMA / barrier sequence, no arithmetic
                for (int c = 0; c < NC1; ++c) {
                    char* nxt = buf ? wt0 : wt1;
                    if (c % CPT == 0) { if (c + CPT < NC1) dma_w2(c / CPT + 1, nxt); else dma_w3(0, nxt); }
                    if (c % CPT == CPT - 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); buf ^= 1; }
                }
                for (int u = 0; u < T3; ++u) {
                    char* nxt = buf ? wt0 : wt1;
                    if (u % UPT == 0) { if (u + UPT < T3) dma_w3(u / UPT + 1, nxt); else if (strip + 1 < nstrips) dma_w2(0, nxt); }
                    if (u % UPT == UPT - 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); buf ^= 1; }
                }
                continue;
            }
        }
        float dx = 0.f, dy = 0.f, dz = 0.f;
        // a lane's j4-th float4 of a 32-channel chunk: channels 8 j4 + 4 half + (0..3) -- the D layout of a 32 x 32 MFMA, so that layer 1
        // may come from the matrix pipe (BF16, F16X2 feature mode) or from the VALU without changing the W2 images (W2PERM)
        auto qi = [&](int j4) { return 2 * j4 + half; };
        const float4* prow = nullptr;
        const float4* trow[3] = {nullptr, nullptr, nullptr};       // ROWS: the three table rows of this lane's point ...
        float tw[3] = {0.f, 0.f, 0.f};                             // ... and their inverse-distance weights
        u32x4 b1f = {0u, 0u, 0u, 0u};                      // L1M / L1F: the B operand of the layer-1 MFMA
        u32x4 b1g = {0u, 0u, 0u, 0u};                      // L1F, BF16X3: the second one ([x0 | x2])
        float cj = 1.f;                                    // L1F: s1 / kappa_j of this lane's neighbour
        // ROWS: raw = (w0 T0 + w1 T1) + w2 T2 of chunk c (pointnet2_utils.py:303 applied to the layer-1 table; the table is stored
        // scaled by s1 in F16X2, so the blend is s1 H1 before the ReLU)
        f32x4 trw[ROWS ? 3 : 1][4];
        auto fetch = [&](int c) {                   // issue the loads of chunk c's three table rows
#pragma unroll
            for (int j = 0; j < ((ROWS && !DIRECT) ? 3 : 1); ++j)
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    if constexpr ((NS == 1 || NS == 4) && DIRECT) {
                        if (p.t_bf16) {          // 4 bf16 / fp16 values = 8 bytes at value offset 32 c + 4 qi(j4) of this lane's row
                            const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(trow[j]) + (size_t)(c * 32 + 4 * qi(j4)) * 2);
                            if constexpr (NS == 1) {
                                trw[j][j4] = f32x4{__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16),
                                                   __uint_as_float(h.y & 0xffff0000u)};
                            } else {             // widened exactly; finish_slice converts back to the same fp16 values (they ARE the plane)
                                const f16x2 a = __builtin_bit_cast(f16x2, h.x), b_ = __builtin_bit_cast(f16x2, h.y);
                                trw[j][j4] = f32x4{(float)a[0], (float)a[1], (float)b_[0], (float)b_[1]};
                            }
                            continue;
                        }
                    }
                    trw[j][j4] = *reinterpret_cast<const f32x4*>(trow[j] + c * 8 + qi(j4));
                }
        };
        auto blend = [&]() {
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    raw[j4][e] = DIRECT ? ((F16 && !(NS == 4 && p.t_bf16)) ? trw[0][j4][e] * s1 : trw[0][j4][e])       // (fp16 rows arrive scaled)
                                        : __fmaf_rn(tw[2], trw[ROWS ? 2 : 0][j4][e], __fmaf_rn(tw[1], trw[ROWS ? 1 : 0][j4][e], __fmul_rn(tw[0], trw[0][j4][e])));
        };
        if constexpr (DIRECT) {
            trow[0] = ((NS == 1 || NS == 4) && p.t_bf16)
                          ? reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.P1) + ((size_t)b * p.N + min(row0 + l31, p.N - 1)) * p.ldp * 2)
                          : reinterpret_cast<const float4*>(p.P1 + ((size_t)b * p.N + min(row0 + l31, p.N - 1)) * p.ldp);
            fetch(0);
            blend();
        } else if constexpr (ROWS) {
            const size_t gr = ((size_t)b * p.N + min(row0 + l31, p.N - 1)) * 3;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                trow[j] = reinterpret_cast<const float4*>(p.P1 + ((size_t)b * p.Npts + p.nn_idx[gr + j]) * p.ldp);
                tw[j] = p.nn_w[gr + j];
            }
            fetch(0);
            blend();
        } else {
            const int idx = XPF ? idx_cur : gi[strip * 32 + l31];
            const float4 q = XPF ? q_cur : p.pts4[(size_t)b * p.Npts + idx];
            if constexpr (XPF) idx_nxt = (strip + 1 < my_strips) ? gi[(strip + 1) * 32 + l31] : idx_cur;
            dx = __fsub_rn(q.x, ctr.x); dy = __fsub_rn(q.y, ctr.y); dz = __fsub_rn(q.z, ctr.z);
            if constexpr (F16 && !L1M) { if (!fmode) { dx *= s1; dy *= s1; dz *= s1; } }      // exact; with P1' = s1 P1 this makes layer 1 produce s1 H1
            if (fmode) {
                if constexpr (L1F && NS == 3) {
                    // (no XPF in this mode: the feature row is requested here)  B1f = [x0 | x1], B1g = [x0 | x2] of v = (f0..f4, dx, dy, dz)
                    const float4* fr = reinterpret_cast<const float4*>(p.feat + ((size_t)b * p.Npts + idx) * p.ldf);
                    const float4 fa = fr[0], fb = fr[1];
                    const float v[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, dx, dy, dz};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        unsigned o[3];
                        split_planes<3>(v[2 * w], v[2 * w + 1], o);
                        b1f[w] = half ? o[1] : o[0];
                        b1g[w] = half ? o[2] : o[0];
                    }
                }
                if constexpr (L1F && F16) {
                    // B1 = [xh(v0..v7) | xl(v0..v7)], v = (f0..f4, dx, dy, dz) of this lane's neighbour times its own power of two s_j
                    float v[8] = {f0_cur.x, f0_cur.y, f0_cur.z, f0_cur.w, f1_cur.x, dx, dy, dz};
                    const float af = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), fabsf(v[4]));
                    const float ax = fmaxf(fmaxf(fabsf(v[5]), fabsf(v[6])), fabsf(v[7]));
                    const float kap = fminf(f16x2_scale(__float_as_uint(af)) * (1.f / p.u1f), f16x2_scale(__float_as_uint(ax)) * (1.f / p.u1x));
                    const float sgf = kap * p.u1f, sgx = kap * p.u1x;
                    cj = s1 * pow2_inverse(kap);
#pragma unroll
                    for (int j = 0; j < 5; ++j) v[j] *= sgf;
#pragma unroll
                    for (int j = 5; j < 8; ++j) v[j] *= sgx;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        unsigned o[2];
                        split_planes<2>(v[2 * w], v[2 * w + 1], o);
                        b1f[w] = half ? o[1] : o[0];
                    }
                }
            } else if constexpr (L1M) {
                // B1: this lane's 8 k slots of its neighbour (half 0: k 0..7, half 1: k 8..15), hi / lo bf16 planes of the inputs
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = 0.f;
                if (hasfeat) {
                    f[0] = f0_cur.x; f[1] = f0_cur.y; f[2] = f0_cur.z; f[3] = f0_cur.w; f[4] = f1_cur.x;
                    if constexpr (F16) {
#pragma unroll
                        for (int j = 0; j < 5; ++j) f[j] *= b1s_f;          // exact (power of two); < 2^8 by the choice of s1 and a1f
                    }
                } else {
                    prow = reinterpret_cast<const float4*>(p.P1 + ((size_t)b * p.Npts + idx) * p.ldp);
                    if (strip == 0) {          // (later strips: requested during the previous strip's layer 3)
#pragma unroll
                        for (int j4 = 0; j4 < 4; ++j4) raw[j4] = *reinterpret_cast<const f32x4*>(prow + qi(j4));
                    }
                }
                // x minus its high 16-bit plane (bf16 / fp16, round to nearest even): exact
                auto lo_of = [](float x) {
                    if constexpr (F16) return x - (float)(_Float16)x;
                    else return x - __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x, 0.f}, bf16x2)) << 16);
                };
                const float ex = F16 ? dx * b1s_x : dx, ey = F16 ? dy * b1s_x : dy, ez = F16 ? dz * b1s_x : dz;
                const float va[8] = {f[0], f[1], f[2], f[3], f[4], lo_of(ex), lo_of(ey), lo_of(ez)};
                // (the constant slot that carries b1: unused in table mode -- its A entry is 0 -- and there s1 alone may exceed fp16: 0 x inf)
                const float vb[8] = {ex, ey, ez, F16 ? (hasfeat ? b1s_b : 0.f) : 1.f, lo_of(f[0]), lo_of(f[1]), lo_of(f[2]), lo_of(f[3])};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    unsigned o[1];
                    split_planes<NS>(half ? vb[2 * w] : va[2 * w], half ? vb[2 * w + 1] : va[2 * w + 1], o);
                    b1f[w] = o[0];
                }
            } else {
                prow = reinterpret_cast<const float4*>(p.P1 + ((size_t)b * p.Npts + idx) * p.ldp);
                if (!XPF || strip == 0) {
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) raw[j4] = *reinterpret_cast<const f32x4*>(prow + qi(j4));
                }
            }
        }
        // layer 1 of chunk c on the matrix pipe.  BF16: one MFMA (C = the table row in table mode); F16X2 feature mode: the two
        // products with [wl | 0] and [wh | wh]
        auto layer1 = [&](int c) {
            f32x16 acc;
            if constexpr (L1F && NS == 3) {
                // C = b1: D register 4q + e of a lane is channel 32c + 8q + 4 half + e
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(sb1w + 32 * c + 8 * q + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * q + e] = bv[e];
                }
                const char* ar = smem + WBYTES + (32 * c + l31) * Cfg::RSA + half * 16;
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(ar), a1 = *reinterpret_cast<const u32x4*>(ar + 32), a2 = *reinterpret_cast<const u32x4*>(ar + 64);
                acc = mfma_planes<3>(a2, b1g, acc);
                acc = mfma_planes<3>(a1, b1f, acc);
                return mfma_planes<3>(a0, b1f, acc);
            } else if constexpr (L1F) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const char* ar = smem + WBYTES + (32 * c + l31) * Cfg::RSA + half * 16;
                const u32x4 ah = *reinterpret_cast<const u32x4*>(ar), al = *reinterpret_cast<const u32x4*>(ar + 32);
                acc = mfma_planes<2>(al, b1f, acc);
                return mfma_planes<2>(ah, b1f, acc);
            } else {
            if (hasfeat) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = raw[r >> 2][r & 3];
            }
            const u32x4 a1 = *reinterpret_cast<const u32x4*>(smem + WBYTES + (32 * c + l31) * 32 + half * 16);
            return mfma_planes<NS>(a1, b1f, acc);
            }
        };
        f32x16 d1;
        if constexpr (L1F) { if (fmode) d1 = layer1(0); }
        if constexpr (L1M) {
            d1 = layer1(0);
            if (!hasfeat && NC1 > 1) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) raw[j4] = *reinterpret_cast<const f32x4*>(prow + 8 + qi(j4));
            }
        }

        // ---------------- layer 2 (contraction-chunk outer): h2[t] = D2[channel 32t + mfma_row(r,half)][neighbour]
        // (the accumulators start at the b2 bias: D rows 4j..4j+3 of a lane are 4 consecutive channels)
        f32x16 h2[T2];
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(sbw + 32 * t + 8 * j + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) h2[t][4 * j + e] = bv[e];
            }

#ifdef EV2H_SAB_TIMELINE
        if (dbgw && strip == dbg_strip) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        STAMP(1);
        // layer-1 finish in fp32, then split, of SLICE j4 of chunk c: the lane's channels 32c + 8 j4 + 4 half + [0, 4) = two of its 16
        // k-slots of the chunk (a chunk's 16 slots feed 2 MFMAs per tile: bp[k-block][plane])
        auto finish_slice = [&](int c, int j4, u32x4 (&bp)[2][NPL]) {
            if constexpr (L1M) {
                // ReLU on the packed bf16 / fp16 pairs: one v_pk_max_i16 per pair (negative floats are negative int16 patterns in both
                // formats).  F16: D1 is s1 H1 already (the factors rode on the operands).  No clamp of an overflowed conversion (the
                // two-plane mode's relu_sat_f16 is free, here it would be a v_pk_min_i16 per pair): see F16_CLAMP
#pragma unroll
                for (int w = 2 * j4; w < 2 * j4 + 2; ++w) {
                    unsigned o[1];
                    split_planes<NS>(d1[2 * w], d1[2 * w + 1], o);
                    bp[w >> 2][0][w & 3] = (F16 && F16_CLAMP) ? sat_pk_f16(relu_pk_bf16(o[0])) : relu_pk_bf16(o[0]);
                }
            } else if (fmode) {
                if constexpr (L1F && NS == 3) {
                    // H1 = relu(D1) (the bias went in as C): D register 4q + e of a lane is channel 32c + 8q + 4 half + e
                    const int q = j4;
                    unsigned lo[3], hi[3];
                    split_planes<3>(relu_bits(d1[4 * q]), relu_bits(d1[4 * q + 1]), lo);
                    split_planes<3>(relu_bits(d1[4 * q + 2]), relu_bits(d1[4 * q + 3]), hi);
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) {
                        bp[q >> 1][s_][(q & 1) * 2 + 0] = lo[s_];
                        bp[q >> 1][s_][(q & 1) * 2 + 1] = hi[s_];
                    }
                }
                if constexpr (L1F && F16) {
                    // s1 H1 = relu(D1 (u1 s1 / s_j) + s1 b1): D register 4q + e of a lane is channel 32c + 8q + 4 half + e
                    const int q = j4;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(sb1w + 32 * c + 8 * q + 4 * half);
                    unsigned lo[NPL], hi[NPL];
                    split_planes<NS>(relu_sat_f16(__fmaf_rn(d1[4 * q], cj, bv[0])), relu_sat_f16(__fmaf_rn(d1[4 * q + 1], cj, bv[1])), lo);
                    split_planes<NS>(relu_sat_f16(__fmaf_rn(d1[4 * q + 2], cj, bv[2])), relu_sat_f16(__fmaf_rn(d1[4 * q + 3], cj, bv[3])), hi);
#pragma unroll
                    for (int s_ = 0; s_ < NPL; ++s_) {
                        bp[q >> 1][s_][(q & 1) * 2 + 0] = lo[s_];
                        bp[q >> 1][s_][(q & 1) * 2 + 1] = hi[s_];
                    }
                }
            } else {
                const f32x4* wp = sW1xT + (8 * c + 2 * j4 + half) * 3;
                const f32x4 wx = wp[0], wy = wp[1], wz = wp[2];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = ROWS ? raw[j4][e] : __fmaf_rn(wz[e], dz, __fmaf_rn(wy[e], dy, __fmaf_rn(wx[e], dx, raw[j4][e])));
                unsigned lo[NPL], hi[NPL];
                if constexpr (DIRECT) {                  // the rows are the layer's input as it is (|v| < 2^15 by the scale)
                    split_planes<NS>(v[0], v[1], lo);
                    split_planes<NS>(v[2], v[3], hi);
                } else if constexpr (NS == 1) {
                    split_planes<NS>(v[0], v[1], lo);
                    split_planes<NS>(v[2], v[3], hi);
                    lo[0] = relu_pk_bf16(lo[0]); hi[0] = relu_pk_bf16(hi[0]);
                } else if constexpr (F16) {
                    split_planes<NS>(relu_sat_f16(v[0]), relu_sat_f16(v[1]), lo);
                    split_planes<NS>(relu_sat_f16(v[2]), relu_sat_f16(v[3]), hi);
                } else {
                    split_planes<NS>(relu_bits(v[0]), relu_bits(v[1]), lo);
                    split_planes<NS>(relu_bits(v[2]), relu_bits(v[3]), hi);
                }
#pragma unroll
                for (int s = 0; s < NPL; ++s) {
                    bp[j4 >> 1][s][(j4 & 1) * 2 + 0] = lo[s];
                    bp[j4 >> 1][s][(j4 & 1) * 2 + 1] = hi[s];
                }
            }
        };
        // the MFMAs of one chunk: the 2*T2 (k-block m, tile t) groups are taken two at a time and their MFMAs alternate between the
        // two tiles' accumulators: anything issued between two MFMAs on the SAME accumulator (here the next fragment reads) costs
        // ~43 cycles, between MFMAs on different accumulators ~6 (MI355X_MICROARCH.md, latency table)
        // FRAG_PIPE (BF16): the fragment reads of group pr + 1 are issued BEFORE the MFMAs of group pr (two register sets, order
        // pinned with scheduling barriers).  Left to itself the compiler emitted read, read, s_waitcnt lgkmcnt(0), MFMA, MFMA per
        // group -- the full LDS latency in front of every MFMA pair, which with ONE product per operand pair is most of a tile
        // step (phase timeline, profiles/r4_sa_timeline.txt: 1.0-1.7 us per 14-MFMA chunk whose MFMAs take 0.22 us).
        // between(pr) runs after the MFMAs of pair pr were issued (L2PIPE: slices of the next chunk's layer-1 finish).
        auto mfma_groups = [&](const char* cur, u32x4 (&bp)[2][NPL], auto&& between) {
            const char* pa = cur + l31 * RS2 + (16 * half) * 2;
            auto ld2 = [&](int pr, u32x4 (&x0)[NPL], u32x4 (&x1)[NPL]) {
                const int m0 = (2 * pr) / T2, t0 = (2 * pr) % T2, m1 = (2 * pr + 1) / T2, t1 = (2 * pr + 1) % T2;
#pragma unroll
                for (int s = 0; s < NPL; ++s) {
                    x0[s] = *reinterpret_cast<const u32x4*>(pa + 32 * t0 * RS2 + s * 64 + m0 * 16);
                    x1[s] = *reinterpret_cast<const u32x4*>(pa + 32 * t1 * RS2 + s * 64 + m1 * 16);
                }
            };
            u32x4 fa[2][2][NPL];
            if constexpr (FRAG_PIPE) ld2(0, fa[0][0], fa[0][1]);
#pragma unroll
            for (int pr = 0; pr < T2; ++pr) {
                const int m0 = (2 * pr) / T2, t0 = (2 * pr) % T2, m1 = (2 * pr + 1) / T2, t1 = (2 * pr + 1) % T2;
                if constexpr (FRAG_PIPE) {
                    if (pr + 1 < T2) ld2(pr + 1, fa[(pr + 1) & 1][0], fa[(pr + 1) & 1][1]);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    ld2(pr, fa[pr & 1][0], fa[pr & 1][1]);
                }
                u32x4 (&a0)[NPL] = fa[pr & 1][0];
                u32x4 (&a1)[NPL] = fa[pr & 1][1];
#pragma unroll
                for (int j = 0; j < PL::NPROD; ++j) {
                    // PACK4: the last tile's high-plane image carries its low plane in rows 8..11 (see SaBCfg)
                    if (!(Cfg::PACK4 && t0 == T2 - 1 && PL::A[j] == 1)) h2[t0] = mfma_planes<NS>(a0[PL::A[j]], bp[m0][PL::B[j]], h2[t0]);
                    if (!(Cfg::PACK4 && t1 == T2 - 1 && PL::A[j] == 1)) h2[t1] = mfma_planes<NS>(a1[PL::A[j]], bp[m1][PL::B[j]], h2[t1]);
                }
                if constexpr (FRAG_PIPE) __builtin_amdgcn_sched_barrier(0);
                between(pr);
            }
        };
        auto l2_dma = [&](int c) {                      // the next tile step's image into the other buffer
            if constexpr (!RES) {
                char* nxt = buf ? wt0 : wt1;
                if (c % CPT == 0) { if (c + CPT < NC1) dma_w2(c / CPT + 1, nxt); else dma_w3(0, nxt); }
            }
        };
        auto l2_sync = [&](int c) {
            STAMP(4 + 4 * c);
            if constexpr (!RES) {
                if (c % CPT == CPT - 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile DMA issued at the start of the step has landed
                    __syncthreads();
                    buf ^= 1;
                }
            }
            STAMP(5 + 4 * c);
        };
        u32x4 bpA[2][NPL];
        if constexpr (!L2PIPE) {
#pragma unroll 1
            for (int c = 0; c < NC1; ++c) {
                STAMP(2 + 4 * c);
                char* cur = RES ? smem + c * Cfg::TB2 : (buf ? wt1 : wt0) + (c % CPT) * Cfg::TB2;
                l2_dma(c);
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) finish_slice(c, j4, bpA);
                if constexpr (ROWS) {
                    if (c + 1 < NC1) fetch(c + 1);      // in flight under this chunk's MFMAs, blended after them
                } else if (!L1M && !fmode && c + 1 < NC1) {
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) raw[j4] = *reinterpret_cast<const f32x4*>(prow + (c + 1) * 8 + qi(j4));
                }
                STAMP(3 + 4 * c);
                mfma_groups(cur, bpA, [](int) {});
                if constexpr (ROWS) {
                    if (c + 1 < NC1) blend();
                }
                if constexpr (L1F) {
                    if (fmode && c + 1 < NC1) d1 = layer1(c + 1);
                }
                if constexpr (L1M) {
                    // next chunk's layer 1: issued behind this chunk's MFMAs, converted at the top of the next iteration; its table row
                    // (table mode) was loaded one iteration ago, the row after it is requested now
                    if (c + 1 < NC1) {
                        d1 = layer1(c + 1);
                        if (!hasfeat && c + 2 < NC1) {
#pragma unroll
                            for (int j4 = 0; j4 < 4; ++j4) raw[j4] = *reinterpret_cast<const f32x4*>(prow + (c + 2) * 8 + qi(j4));
                        }
                    }
                }
                l2_sync(c);
            }
        } else {
            // L2PIPE: the layer-1 finish + split of chunk c + 1 is cut into its four slices and placed BETWEEN the MFMA pairs of chunk c
            // (two operand buffers), so that one wave's conversion runs under its SIMD partner's MFMAs instead of both waves
            // converting behind the same barrier while the matrix pipe idles (phase timeline: 13 % of an f16x2 strip).
            u32x4 bpB[2][NPL];
            const bool vtab = !L1M && !fmode;            // layer 1 on the VALU from gathered table rows
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                finish_slice(0, j4, bpA);
                if (vtab) raw[j4] = *reinterpret_cast<const f32x4*>(prow + 8 + qi(j4));
            }
            auto l2_step = [&](int c, u32x4 (&bc)[2][NPL], u32x4 (&bn)[2][NPL]) {
                STAMP(2 + 4 * c);
                char* cur = RES ? smem + c * Cfg::TB2 : (buf ? wt1 : wt0) + (c % CPT) * Cfg::TB2;
                l2_dma(c);
                const bool more = c + 1 < NC1;
                if (more) {
                    if constexpr (L1F) { if (fmode) d1 = layer1(c + 1); }
                    if constexpr (L1M) {
                        d1 = layer1(c + 1);
                        if (!hasfeat && c + 2 < NC1) {
#pragma unroll
                            for (int j4 = 0; j4 < 4; ++j4) raw[j4] = *reinterpret_cast<const f32x4*>(prow + (c + 2) * 8 + qi(j4));
                        }
                    }
                }
                STAMP(3 + 4 * c);
                mfma_groups(cur, bc, [&](int pr) {
                    if (more) {
#pragma unroll
                        for (int j4 = 0; j4 < 4; ++j4) {
                            if (j4 * T2 / 4 == pr) {
                                __builtin_amdgcn_sched_barrier(0);
                                finish_slice(c + 1, j4, bn);
                                if (vtab && c + 2 < NC1) raw[j4] = *reinterpret_cast<const f32x4*>(prow + (c + 2) * 8 + qi(j4));
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                });
                l2_sync(c);
            };
#pragma unroll 1
            for (int c = 0; c < NC1; c += 2) {
                l2_step(c, bpA, bpB);
                l2_step(c + 1, bpB, bpA);
            }
        }

        STAMP(38);
        if constexpr (XPF) {
            // next strip's coordinates and feature row / first table chunk: its index arrived long ago; `raw` is free again
            if (strip + 1 < my_strips) {
                q_cur = p.pts4[(size_t)b * p.Npts + idx_nxt];
                if (hasfeat) {
                    const float4* fr = reinterpret_cast<const float4*>(p.feat + ((size_t)b * p.Npts + idx_nxt) * p.ldf);
                    f0_cur = fr[0]; f1_cur = fr[1];
                } else {
                    const float4* pn = reinterpret_cast<const float4*>(p.P1 + ((size_t)b * p.Npts + idx_nxt) * p.ldp);
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) raw[j4] = *reinterpret_cast<const f32x4*>(pn + qi(j4));
                }
                idx_cur = idx_nxt;
            }
        }
        if constexpr (Cfg::PACK4) {          // wl*xh of the leftover channels arrived in D rows 8..11 = registers 4..7
#pragma unroll
            for (int r = 0; r < 4; ++r) { h2[T2 - 1][r] += h2[T2 - 1][r + 4]; h2[T2 - 1][r + 4] = 0.f; }
        }
        // ReLU in fp32 (the bias is already in), then split in place: h2p[s][t][k] packs D2 rows (2k, 2k+1) of tile t
        u32x4 h2p[NPL][T2][2];        // [plane][tile][k-block m]: directly in MFMA A-operand form
        // one k-block (8 channels per lane) of tile t; PACK4: the last tile's block 0 is re-packed once both planes exist
        auto split_half = [&](int t, int m) {
#pragma unroll
            for (int k = 4 * m; k < 4 * m + 4; ++k) {
                unsigned o[NPL];
                if constexpr (C2ONE) {      // (c2 = 1)
                    // cvt + one packed integer max, as in BF16.  No clamp: the bound behind s1 is rigorous, and a caller who breaks the
                    // contract behind it (dmax) gets inf -> NaN outputs instead of silently clamped ones.  (Same-box A/B, whole step,
                    // profiles/r6_f16_chain_scale.txt: a factor per layer 40 630, chain scale 41 200, without the clamps 41 740 windows/s.)
                    split_planes<NS>(h2[t][2 * k], h2[t][2 * k + 1], o);
                    o[0] = F16_CLAMP ? sat_pk_f16(relu_pk_bf16(o[0])) : relu_pk_bf16(o[0]);
                }
                else if constexpr (F16) split_planes<NS>(relu_sat_f16(h2[t][2 * k] * c2), relu_sat_f16(h2[t][2 * k + 1] * c2), o);
                else if constexpr (NS == 1) { split_planes<NS>(h2[t][2 * k], h2[t][2 * k + 1], o); o[0] = relu_pk_bf16(o[0]); }     // (u2 = 1: no factor)
                else split_planes<NS>(relu_bits(h2[t][2 * k] * c2), relu_bits(h2[t][2 * k + 1] * c2), o);
#pragma unroll
                for (int s = 0; s < NPL; ++s) h2p[s][t][k >> 2][k & 3] = o[s];
            }
            if constexpr (Cfg::PACK4) {
                if (t == T2 - 1 && m == 0) {
                    // last k-block: slots [xh(4) | xl(4)] in the lower half-wave, [xh(4) | 0] in the upper one (which gets the values from
                    // its partner lane: same neighbour, other half); the packer stores [wh | wh | wl | 0] at these positions of the W3 image
                    const unsigned h01 = h2p[0][T2 - 1][0][0], h23 = h2p[0][T2 - 1][0][1];
                    const unsigned l01 = h2p[1][T2 - 1][0][0], l23 = h2p[1][T2 - 1][0][1];
                    const unsigned ph01 = (unsigned)__shfl_xor((int)h01, 32, 64), ph23 = (unsigned)__shfl_xor((int)h23, 32, 64);
                    h2p[0][T2 - 1][0][0] = half ? ph01 : h01;
                    h2p[0][T2 - 1][0][1] = half ? ph23 : h23;
                    h2p[0][T2 - 1][0][2] = half ? 0u : l01;
                    h2p[0][T2 - 1][0][3] = half ? 0u : l23;
                }
            }
        };
        // H2FUSE: only tile 0 is split here; tile t + 1 is split between the MFMA groups of tile t in the FIRST layer-3 step, so that
        // the conversion (VALU) of one wave runs under the MFMAs of its SIMD partner instead of both waves converting while the matrix
        // pipe idles (phase timeline: "h2 ReLU + split" was 8 % of a strip).  Same values, same order of the contraction.
        // L3T16: regroup the 16-lane rows of tile t's two k-block registers into the A operands of the two neighbour tiles
        auto perm16 = [&](int t) {
            if constexpr (L3T16) {
#pragma unroll
                for (int s = 0; s < NPL; ++s)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                        const u32x2_ a = __builtin_amdgcn_permlane32_swap(h2p[s][t][0][w], h2p[s][t][1][w], false, false);
                        const u32x2_ b_ = __builtin_amdgcn_permlane16_swap(a[0], a[1], false, false);
                        h2p[s][t][0][w] = b_[0];          // neighbours 0..15, k-groups 0..3 of tile t
                        h2p[s][t][1][w] = b_[1];          // neighbours 16..31
                    }
            }
        };
#pragma unroll
        for (int t = 0; t < (H2FUSE ? 1 : T2); ++t) { split_half(t, 0); split_half(t, 1); perm16(t); }
        STAMP(39);

        // ---------------- layer 3 + max: D3[neighbour][channel] = H2 (A, registers) x W3 tile (B, LDS, permuted k order)
        // set abstraction: the max over the strip's neighbours, kept per output tile
        auto finish_tile = [&](int u, const f32x16& acc) {
            float mx = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, acc[r]), acc[r + 1]);
            mx = fmaxf(mx, acc[15]);
#pragma unroll
            for (int uu = 0; uu < T3; ++uu) mrun[uu] = (uu == u) ? fmaxf(mrun[uu], mx) : mrun[uu];
        };
        // TPS = 2 (two tiles resident per step: every MLP but 32-32-64): the two tiles' accumulators take alternate MFMAs (consecutive
        // MFMAs never depend on each other) and share the activation fragments; TPS = 1: one tile, two accumulators that are summed
        // (the row chains keep one tile at a time: their store addresses would not fit the scalar registers twice)
        constexpr int TPS = (!ROWS && (RES || UPT == 2)) ? 2 : 1;
        static_assert(T3 % TPS == 0, "layer-3 tiles are walked in pairs");
        auto l3_step = [&](int u, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            STAMP(40 + 4 * u);
            char* cur = RES ? smem + NC1 * Cfg::TB2 + u * Cfg::TB3 : (buf ? wt1 : wt0) + (u % UPT) * Cfg::TB3;
            char* nxt = buf ? wt0 : wt1;
            if constexpr (!RES) {
                if (u % UPT == 0) {
                    const bool more_w3 = (u + UPT < T3);
                    const bool more = more_w3 || (strip + 1 < nstrips);
                    if (more_w3) dma_w3(u / UPT + 1, nxt); else if (more) dma_w2(0, nxt);
                }
            }
            float b3u = 0.f;
            if constexpr (ROWS) b3u = p.b3[32 * u + l31];      // (requested here, used after the step's MFMAs)
            if constexpr (L3T16) {
                // two output tiles (u, u + 1) x two neighbour tiles x two 16-channel halves: eight 16 x 16 accumulators (32 registers,
                // as the two 32 x 32 ones); groups (t, c): one B fragment per plane and output tile feeds both neighbour tiles
                f32x4 a16[2][2][2];          // [output tile][neighbour tile][channel half]
#pragma unroll
                for (int i = 0; i < 8; ++i) a16[i >> 2][(i >> 1) & 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int lg = lane >> 4, l15 = lane & 15;
                const char* pb16 = cur + l15 * RS3;
                constexpr int NG16 = 2 * T2;
                auto ld16 = [&](int g, u32x4 (&x)[2][NPL]) {
                    const int t = g >> 1, c = g & 1;
                    // the last tile of a width that fills only its first k-block (M_LAST == 1): k-groups 2, 3 do not exist in the image
                    // (their A values are exact zeros: padded channels) -- read k-groups 0, 1 again instead of whatever follows the row
                    const int kg = (t == T2 - 1 && Cfg::M_LAST == 1) ? (lg & 1) : lg;
#pragma unroll
                    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                        for (int s = 0; s < NPL; ++s)
                            x[uu][s] = *reinterpret_cast<const u32x4*>(pb16 + uu * Cfg::TB3 + (16 * c) * RS3 + s * (C2P * 2) + (32 * t + 8 * kg) * 2);
                };
                u32x4 fw16[2][2][NPL];
                if constexpr (FRAG_PIPE) ld16(0, fw16[0]);
#pragma unroll
                for (int g = 0; g < NG16; ++g) {
                    const int t = g >> 1, c = g & 1;
                    if constexpr (FRAG_PIPE) {
                        if (g + 1 < NG16) ld16(g + 1, fw16[(g + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        ld16(g, fw16[g & 1]);
                    }
                    u32x4 (&w)[2][NPL] = fw16[g & 1];
#pragma unroll
                    for (int j = 0; j < PL::NPROD; ++j) {
                        if (Cfg::PACK4 && t == T2 - 1 && !(PL::A[j] == 0 && PL::B[j] == 0)) continue;   // one MFMA holds all three products
#pragma unroll
                        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                            for (int nu = 0; nu < 2; ++nu)
                                a16[uu][nu][c] = mfma16_planes<NS>(h2p[PL::A[j]][t][nu], w[uu][PL::B[j]], a16[uu][nu][c]);
                    }
                    if constexpr (FRAG_PIPE) __builtin_amdgcn_sched_barrier(0);
                    if constexpr (FIRST) {
                        if (t + 1 < T2) {
                            if constexpr (!FRAG_PIPE) __builtin_amdgcn_sched_barrier(0);
                            split_half(t + 1, c);                      // tile t + 1 is converted between the groups of tile t ...
                            if (c == 1) perm16(t + 1);                 // ... and regrouped once both of its k-blocks exist
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                STAMP(41 + 4 * u);
                // max over this lane's 8 rows (2 neighbour tiles x 4 registers) per (output tile, channel half)
#pragma unroll
                for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const f32x4 x = a16[uu][0][c], y = a16[uu][1][c];
                        const float mx = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])));
#pragma unroll
                        for (int q = 0; q < T3; ++q) m16[q][c] = (q == u + uu) ? fmaxf(m16[q][c], mx) : m16[q][c];
                    }
            } else {
            f32x16 acc, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
            const char* pb = cur + l31 * RS3 + (8 * half) * 2;
            constexpr int NG3 = 2 * (T2 - 1) + Cfg::M_LAST;       // live (tile, k-block) groups of the contraction
            auto ld3 = [&](int g, u32x4 (&x)[NPL], u32x4 (&x1)[TPS == 2 ? NPL : 1]) {
#pragma unroll
                for (int s = 0; s < NPL; ++s) {
                    x[s] = *reinterpret_cast<const u32x4*>(pb + s * (C2P * 2) + (16 * g) * 2);
                    if constexpr (TPS == 2) x1[s] = *reinterpret_cast<const u32x4*>(pb + Cfg::TB3 + s * (C2P * 2) + (16 * g) * 2);
                }
            };
            u32x4 fw[2][NPL], fw1[2][TPS == 2 ? NPL : 1];
            if constexpr (FRAG_PIPE) ld3(0, fw[0], fw1[0]);
#pragma unroll
            for (int t = 0; t < T2; ++t) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (t < T2 - 1 || m < Cfg::M_LAST) {
                        const int g = 2 * t + m;
                        if constexpr (FRAG_PIPE) {
                            if (g + 1 < NG3) ld3(g + 1, fw[(g + 1) & 1], fw1[(g + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                        } else {
                            ld3(g, fw[g & 1], fw1[g & 1]);
                        }
                        u32x4 a[NPL];
                        u32x4 (&w)[NPL] = fw[g & 1];
                        u32x4 (&w1)[TPS == 2 ? NPL : 1] = fw1[g & 1];
#pragma unroll
                        for (int s = 0; s < NPL; ++s) a[s] = h2p[s][t][m];
                        // operand roles swapped w.r.t. layer 2: activations are A, weights are B
#pragma unroll
                        for (int j = 0; j < PL::NPROD; ++j) {
                            if (Cfg::PACK4 && t == T2 - 1 && !(PL::A[j] == 0 && PL::B[j] == 0)) continue;   // one MFMA holds all three products
                            if constexpr (TPS == 2) {
                                acc = mfma_planes<NS>(a[PL::A[j]], w[PL::B[j]], acc);
                                acc1 = mfma_planes<NS>(a[PL::A[j]], w1[PL::B[j]], acc1);
                            } else {
                                if (((2 * t + m) * PL::NPROD + j) & 1) acc1 = mfma_planes<NS>(a[PL::A[j]], w[PL::B[j]], acc1);
                                else acc = mfma_planes<NS>(a[PL::A[j]], w[PL::B[j]], acc);
                            }
                        }
                        if constexpr (FRAG_PIPE) __builtin_amdgcn_sched_barrier(0);
                        if constexpr (FIRST) {
                            if (t + 1 < T2) {
                                if constexpr (!FRAG_PIPE) __builtin_amdgcn_sched_barrier(0);
                                split_half(t + 1, m);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
            }
            STAMP(41 + 4 * u);
            if constexpr (TPS == 2) {
                finish_tile(u, acc);
                finish_tile(u + 1, acc1);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
                if constexpr (ROWS) {
                    // D3[point][channel]: this lane holds channel 32u + l31 of the points 8(r/4) + 4 half + r%4 -- one store
                    // instruction writes two 128-byte row segments
                    float* orow = p.out + ((size_t)b * p.N + row0 + 4 * half) * p.ldo + 32 * u + l31;
                    const bool colok = valid && (32 * u + l31 < p.ncols);
                    float* ocm = p.out_cm ? p.out_cm + (size_t)b * p.out_cm_stride + (size_t)(32 * u + l31) * p.N + row0 + 4 * half : nullptr;
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pt = 8 * (r >> 2) + (r & 3);
                        float o = acc[r] * c3 + b3u;
                        if (p.relu_out) o = fmaxf(o, 0.f);
                        if (colok && row0 + 4 * half + pt < p.N) {
                            if (NS == 1 && p.out_bf16) {
                                unsigned short* o16 = reinterpret_cast<unsigned short*>(p.out) + ((size_t)b * p.N + row0 + 4 * half + pt) * p.ldo + 32 * u + l31;
                                *o16 = (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{o, 0.f}, bf16x2)) & 0xffffu);
                            } else if (NS == 4 && p.out_bf16) {
                                o = fminf(o * so, 65504.f);              // (exact power of two; the bound keeps it below 2^15 -- the clamp is for a broken contract)
                                unsigned short* o16 = reinterpret_cast<unsigned short*>(p.out) + ((size_t)b * p.N + row0 + 4 * half + pt) * p.ldo + 32 * u + l31;
                                const _Float16 h16 = (_Float16)o;
                                *o16 = __builtin_bit_cast(unsigned short, h16);
                                o = (float)h16;                          // the record is of the STORED values
                            } else
                            orow[(size_t)pt * p.ldo] = o;
                            if (ocm) ocm[pt] = o;
                            am = max(am, abs_bits(o));
                        }
                    }
                } else {
                    finish_tile(u, acc);
                }
            }
            }       // (!L3T16)
            STAMP(42 + 4 * u);
            if constexpr (!RES) {
                if ((u + TPS - 1) % UPT == UPT - 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    buf ^= 1;
                }
            }
            STAMP(43 + 4 * u);
        };
        if constexpr (H2FUSE) l3_step(0, std::true_type{});
#pragma unroll 1
        for (int u = H2FUSE ? TPS : 0; u < T3; u += TPS) l3_step(u, std::false_type{});
        STAMP(37);
    }
    if constexpr (L3T16) {
        // the maxima over the lane groups (rows 4 G + r of both neighbour tiles), then this lane's channel 32 u + (l & 31): its half is
        // (l >> 4) & 1 -- from here on the epilogue is the 32 x 32 form's (every lane group holds the same reduced values)
#pragma unroll
        for (int u = 0; u < T3; ++u) {
            float v0 = m16[u][0], v1 = m16[u][1];
            v0 = fmaxf(v0, __shfl_xor(v0, 16, 64)); v1 = fmaxf(v1, __shfl_xor(v1, 16, 64));
            v0 = fmaxf(v0, __shfl_xor(v0, 32, 64)); v1 = fmaxf(v1, __shfl_xor(v1, 32, 64));
            mrun[u] = ((lane >> 4) & 1) ? v1 : v0;
        }
    }

    if constexpr (!ROWS && !RES) {
        if (spg > 1) {
            // combine the strips' partial maxima: [wave][C3] floats in the (now idle) tile buffer; the group's first wave finishes
            float* smax = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int u = 0; u < T3; ++u) {
                const float v = fmaxf(mrun[u], __shfl_xor(mrun[u], 32, 64));
                if (half == 0) smax[wave * C3 + 32 * u + l31] = v;
            }
            __syncthreads();
            if (sw == 0) {
#pragma unroll
                for (int u = 0; u < T3; ++u) {
                    float v = smax[wave * C3 + 32 * u + l31];
                    for (int k = 1; k < spg; ++k) v = fmaxf(v, smax[(wave + k) * C3 + 32 * u + l31]);
                    mrun[u] = v;
                }
            }
        }
    }
    if constexpr (!ROWS) {
#pragma unroll
        for (int u = 0; u < T3; ++u) {
            const float v = fmaxf(mrun[u], __shfl_xor(mrun[u], 32, 64));
            if (valid && half == 0 && sw == 0) {
                float o = fmaxf(v * c3 + p.b3[32 * u + l31], 0.f);
                if constexpr (C2ONE) o = (c3 == c3) ? o : c3;          // (fmaxf drops a NaN: the broken-contract marker must reach the caller)
                p.out[gf * p.ldo + 32 * u + l31] = o;
                am = max(am, __float_as_uint(o));
            }
        }
    }
    if constexpr (!ROWS) {
        if (p.xyz_out && valid && sw == 0 && lane < 8)
            p.xyz_out[gf * p.xyz_ld + lane] = lane == 0 ? ctr.x : lane == 1 ? ctr.y : lane == 2 ? ctr.z : 0.f;
    }
    if constexpr (F16) {
        // Range record of the output (ev2hands_hip.h "Range records"): amax[b] = max over the window's groups.  One device-scope
        // atomicMax per GROUP put 512 read-modify-writes on one address per window and launch; across the 8 XCDs these are
        // executed one after the other at the memory side, and a launch of a few windows (16 windows of 8192 points = one rank's
        // share of BASELINE config 5) spent more time draining them than computing (enc.sa1: 14 -> 99, 44 -> 109 us at 16
        // windows, tools/debug/sa_small_grid.py).  The waves of a workgroup work on consecutive groups -- almost always one
        // window -- so their maxima are combined in LDS first:
        //  * streamed variants (the waves already meet at barriers): one atomic per (workgroup, window);
        //  * resident variant (barrier-free persistent waves, which drift apart -- also across a window boundary): one LDS slot per
        //    window of this workgroup's XCD range holds the workgroup's running maximum of that window; a wave only goes to memory
        //    when it raises it.  (A single (window, max) key dropped the update of a lagging wave once a leading wave had moved
        //    on to the next window -- an under-estimated record, caught by the sharded-equals-unsharded test.)  Windows beyond
        //    the slots (more than 64 windows per XCD) update memory directly.
        // max is exact, associative and idempotent: the record is the same number whatever the route.
        if (p.out_amax) {
            am = wave_max_u32_dpp(am);
            char* rec = smem + WBYTES + Cfg::SMALL_NOREC;
            if constexpr (RES) {
                if (lane == 0 && am) {
                    const int slot = b - (xcd * p.per_xcd) / p.S;          // windows of this XCD's group range, in order
                    unsigned old = 0u;
                    if (slot >= 0 && slot < Cfg::REC_SLOTS) old = atomicMax(reinterpret_cast<unsigned*>(rec) + slot, am);
                    if (am > old) atomicMax(&p.out_amax[b], am);
                }
            } else {
                int* sb = reinterpret_cast<int*>(rec);
                unsigned* sa = reinterpret_cast<unsigned*>(rec + 32);
                if (lane == 0) { sb[wave] = (valid && am) ? b : -1; sa[wave] = am; }
                __syncthreads();
                if (lane == 0 && valid && am) {
                    bool first = true;
                    unsigned m = am;
#pragma unroll
                    for (int w = 0; w < WV; ++w) {
                        if (sb[w] == b) { first = first && w >= wave; m = max(m, sa[w]); }
                    }
                    if (first) atomicMax(&p.out_amax[b], m);
                }
            }
        }
    }
    if constexpr (!RES) break;
  }
}

template <int C1, int C2, int C3, int NS, int MODE = 0>
int launch_sab(SaBP p, hipStream_t st) {
    using Cfg = SaBCfg<C1, C2, C3, NS>;
    static const bool streamed_only = getenv("EV2H_SA_STREAMED") != nullptr;      // A/B switch for the resident variant
    if constexpr (Cfg::FITS_RESIDENT) {
        if (!streamed_only) {
            static PerDevice wg_slot{};                   // per device: 0 = not queried yet, else workgroups per CU
            std::atomic<int>& wg = wg_slot.cur();
            int wg_per_cu = wg.load(std::memory_order_acquire);
            if (!wg_per_cu) {
                auto k = sa_mlp_max_bf16_kernel<C1, C2, C3, NS, true, MODE>;
                EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::RES_LDS_BYTES));
                int n = 0;
                EV2H_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, SAB_THREADS, Cfg::RES_LDS_BYTES));
                wg_per_cu = n > 0 ? n : 1;
                wg.store(wg_per_cu, std::memory_order_release);
            }
            const int ngroups = p.B * p.S;
            const int want = 256 * wg_per_cu;                                   // one resident wave of workgroups
            p.per_xcd = ceil_div(ceil_div(ngroups, 8), SAB_WAVES) * SAB_WAVES;
            p.nblk = 8 * std::min(want / 8, ceil_div(p.per_xcd, SAB_WAVES));
            sa_mlp_max_bf16_kernel<C1, C2, C3, NS, true, MODE><<<p.nblk, SAB_THREADS, Cfg::RES_LDS_BYTES, st>>>(p);
            EV2H_CHECK_LAUNCH();
            return EV2H_OK;
        }
    }
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sa_mlp_max_bf16_kernel<C1, C2, C3, NS, false, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES)););
    {
        // small grids (a few windows at a time): spread a group's strips over the waves of a workgroup (SaBP::spg).  Chosen by the
        // launch size only -- the result does not depend on it (a max is exact and order-free; operator tests at 1 .. 64 windows).
        const int spg = p.K / 32;
        p.spg = 1;
        if ((spg == 2 || spg == 4) && p.nblk < 256 && SAB_WAVES * C3 * 4 <= 2 * Cfg::TILE) {
            p.spg = spg;
            p.nblk = ceil_div(p.B * p.S, SAB_WAVES / spg);
        }
    }
    sa_mlp_max_bf16_kernel<C1, C2, C3, NS, false, MODE><<<p.nblk, SAB_THREADS, Cfg::LDS_BYTES, st>>>(p);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

template <int NS, int MODE = 0>
int dispatch_sab(const SaBP& p, int c1, int c2, int c3, hipStream_t st) {
    if constexpr (sab_split_forms<NS>() && MODE == 0) {
        if (p.feat) return dispatch_sab<NS, 3>(p, c1, c2, c3, st);      // raw feature rows: the feature-row instantiations
    }
    if (c1 == 32 && c2 == 32 && c3 == 64) return launch_sab<32, 32, 64, NS, MODE>(p, st);
    if (c1 == 64 && c2 == 64 && c3 == 128) return launch_sab<64, 64, 128, NS, MODE>(p, st);
    if (c1 == 64 && c2 == 96 && c3 == 128) return launch_sab<64, 96, 128, NS, MODE>(p, st);
    if (c1 == 128 && c2 == 128 && c3 == 256) return launch_sab<128, 128, 256, NS, MODE>(p, st);
    if (c1 == 128 && c2 == 196 && c3 == 256) return launch_sab<128, 196, 256, NS, MODE>(p, st);
    ev2h_set_error("ev2h_sa_mlp_max: unsupported MLP widths %d-%d-%d", c1, c2, c3);
    return EV2H_ERR_ARG;
}

}  // namespace

#ifdef EV2H_SAB_TIMELINE
extern "C" int ev2h_sab_timeline_read(long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sab_timeline), sizeof(long long) * 256);
}
#endif
template <int C1, int C2, int C3, int NS, int MODE>
static int launch_fp(SaBP p, hipStream_t st) {
    using Cfg = SaBCfg<C1, C2, C3, NS>;
    auto k = sa_mlp_max_bf16_kernel<C1, C2, C3, NS, false, MODE>;
    static PerDevice attr_set{};
    EV2H_ONCE_PER_DEVICE(attr_set,
        EV2H_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES)););
    k<<<p.nblk, SAB_THREADS, Cfg::LDS_BYTES, st>>>(p);
    EV2H_CHECK_LAUNCH();
    return EV2H_OK;
}

template <int NS>
static int dispatch_fp(const SaBP& p, const ev2h_fp_desc* d, hipStream_t st) {
    if (d->nn_idx && d->C1 == 128 && d->C2 == 128 && d->C3 == 256) return launch_fp<128, 128, 256, NS, 1>(p, st);
    if (!d->nn_idx && d->C1 == 256 && d->C2 == 256 && d->C3 == 32) return launch_fp<256, 256, 32, NS, 2>(p, st);
    ev2h_set_error("ev2h_fp_mlp: unsupported chain %d-%d-%d (%s)", d->C1, d->C2, d->C3, d->nn_idx ? "interpolated" : "plain rows");
    return EV2H_ERR_ARG;
}

// The tile-image geometry the host packer must reproduce (csrc/pack.hip: sa_images, gemm_image; tests/ref_pack.py): ONE source of
// truth -- both assert their own numbers against this on every pack.
int ev2h_gemm_tile_geometry(int ns, int out[2]);
namespace {
template <int C1, int C2, int C3, int NS>
int fill_geometry(int out[10]) {
    using Cfg = SaBCfg<C1, C2, C3, NS>;
    out[0] = Cfg::T2; out[1] = Cfg::C2P; out[2] = Cfg::RS2; out[3] = Cfg::RS3; out[4] = Cfg::TB2; out[5] = Cfg::TB3;
    out[8] = Cfg::PACK4 ? Cfg::REM : 0;
    out[9] = 1;                        // the W2 images hold their k slots in the D-register order of a 32 x 32 MFMA (see qi / L1M / L1F in the kernel)
    return EV2H_OK;
}
template <int NS>
int geometry_ns(int c1, int c2, int c3, int out[10]) {
    if (c1 == 32 && c2 == 32 && c3 == 64) return fill_geometry<32, 32, 64, NS>(out);
    if (c1 == 64 && c2 == 64 && c3 == 128) return fill_geometry<64, 64, 128, NS>(out);
    if (c1 == 64 && c2 == 96 && c3 == 128) return fill_geometry<64, 96, 128, NS>(out);
    if (c1 == 128 && c2 == 128 && c3 == 256) return fill_geometry<128, 128, 256, NS>(out);
    if (c1 == 128 && c2 == 196 && c3 == 256) return fill_geometry<128, 196, 256, NS>(out);
    if (c1 == 256 && c2 == 256 && c3 == 32) return fill_geometry<256, 256, 32, NS>(out);
    return EV2H_ERR_ARG;
}
}  // namespace

extern "C" int ev2h_tile_geometry(int C1, int C2, int C3, int planes, int out[10]) {
    EV2H_CHECK_ARG(out && planes >= 1 && planes <= 4);        // (4 = the mode code of "f16": one fp16 plane, planes.hpp)
    for (int i = 0; i < 10; ++i) out[i] = 0;
    const int rc = planes == 1 ? geometry_ns<1>(C1, C2, C3, out) : planes == 2 ? geometry_ns<2>(C1, C2, C3, out) : planes == 3 ? geometry_ns<3>(C1, C2, C3, out) : geometry_ns<4>(C1, C2, C3, out);
    if (rc) { ev2h_set_error("ev2h_tile_geometry: unsupported chain %d-%d-%d", C1, C2, C3); return rc; }
    return ev2h_gemm_tile_geometry(planes, out + 6);
}

int ev2h_fp_mlp_ex(const ev2h_fp_desc* d, int t_bf16, int out_bf16, ev2h_stream_t stream, float* row16_scale = nullptr, float w3_norm = 0.f, float b3_max = 0.f);
extern "C" int ev2h_fp_mlp(const ev2h_fp_desc* d, ev2h_stream_t stream) { return ev2h_fp_mlp_ex(d, 0, 0, stream); }

// internal (forward.hip): t_bf16 -- the input rows of form (b) are 16-bit values; out_bf16 -- the output rows are written as 16-bit
// values.  BF16: bf16 values; F16 [r6] (with range records): fp16 values times the per-window power of two row16_scale[b] (written by the
// out_bf16 call from the bound |W3|_1 bound(H2) + max|b3| -- w3_norm, b3_max --, read by the t_bf16 call).
int ev2h_fp_mlp_ex(const ev2h_fp_desc* d, int t_bf16, int out_bf16, ev2h_stream_t stream, float* row16_scale, float w3_norm, float b3_max) {
    EV2H_CHECK_ARG(d && d->T && d->W2s && d->W3s && d->b2 && d->b3 && d->out);
    EV2H_CHECK_ARG(!(t_bf16 || out_bf16) || ((d->precision == EV2H_PREC_BF16 || (d->precision == EV2H_PREC_F16 && row16_scale && d->t_amax)) &&
                                             (!out_bf16 || !d->out_cm) && (!t_bf16 || !d->nn_idx)));
    EV2H_CHECK_ARG((d->nn_idx != nullptr) == (d->nn_w != nullptr));
    EV2H_CHECK_ARG(d->B > 0 && d->N > 0 && d->ldt >= d->C1 && (d->ldt % 4) == 0);
    const int ncols = d->out_cols > 0 ? d->out_cols : d->C3;
    EV2H_CHECK_ARG(ncols <= d->C3 && d->ldo >= ncols);
    if (d->nn_idx) EV2H_CHECK_ARG(d->S >= 3);
    SaBP p{};
    p.P1 = d->T; p.ldp = d->ldt; p.nn_idx = d->nn_idx; p.nn_w = d->nn_w; p.N = d->N;
    p.W2s = (const char*)d->W2s; p.b2 = d->b2; p.W3s = (const char*)d->W3s; p.b3 = d->b3;
    p.out = d->out; p.ldo = d->ldo; p.B = d->B; p.Npts = d->S; p.S = ceil_div(d->N, 32); p.K = 32;
    p.S_total = p.S; p.s_off = 0;
    p.ncols = ncols; p.relu_out = d->no_relu_out ? 0 : 1; p.out_cm = d->out_cm;
    p.t_bf16 = t_bf16; p.out_bf16 = out_bf16;
    p.row16_scale = row16_scale; p.w3_norm = w3_norm; p.b3_max = b3_max;
    p.out_cm_stride = d->out_cm_stride ? d->out_cm_stride : (size_t)ncols * d->N;
    p.u2 = d->w2_unscale > 0.f ? d->w2_unscale : 1.f; p.u3 = d->w3_unscale > 0.f ? d->w3_unscale : 1.f;
    p.nblk = ceil_div(d->B * p.S, SAB_WAVES);
    if (d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_F16) {
        p.out_amax = d->out_amax;
        if (d->t_amax) {
            EV2H_CHECK_ARG(d->w2_norm >= 0.f && d->b2_max >= 0.f);
            EV2H_CHECK_ARG(d->nn_idx ? d->t_scale != nullptr : d->t_scale == nullptr);   // tables arrive scaled, plain rows do not
            // a convex blend of table rows stays inside the table's range (+ rounding): the layer-1 bound is the input's record
            p.p1_scale = d->t_scale; p.p1_amax = d->t_amax; p.w1x_norm = 0.f; p.dmax = 1.f; p.w2_norm = d->w2_norm; p.b2_max = d->b2_max;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (d->precision == EV2H_PREC_BF16X3) return dispatch_fp<3>(p, d, st);
    if (d->precision == EV2H_PREC_F16X2) return dispatch_fp<2>(p, d, st);
    if (d->precision == EV2H_PREC_BF16) return dispatch_fp<1>(p, d, st);
    if (d->precision == EV2H_PREC_F16) return dispatch_fp<4>(p, d, st);
    ev2h_set_error("ev2h_fp_mlp: precision %d is not a 16-bit plane mode (F32: ev2h_three_nn_interp + ev2h_gemm)", d->precision);
    return EV2H_ERR_ARG;
}

// called by ev2h_sa_mlp_max when d->precision != EV2H_PREC_F32
int ev2h_sa_mlp_max_bf16(const ev2h_sa_desc* d, ev2h_stream_t stream) {
    EV2H_CHECK_ARG(d->W2s && d->W3s);
    SaBP p{};
    p.P1 = d->P1; p.ldp = d->ldp; p.pts4 = (const float4*)d->pts4; p.ctr4 = (const float4*)d->ctr4; p.gidx = d->gidx;
    p.W1x = (const float4*)d->W1x; p.W2s = (const char*)d->W2s; p.b2 = d->b2; p.W3s = (const char*)d->W3s; p.b3 = d->b3;
    p.out = d->out; p.ldo = d->ldo; p.B = d->B; p.Npts = d->Npts; p.S = d->S; p.K = d->K;
    p.S_total = d->S_total > 0 ? d->S_total : d->S; p.s_off = d->s_off;
    EV2H_CHECK_ARG(p.s_off >= 0 && p.s_off + p.S <= p.S_total);
    p.cnt = d->cnt; p.cnt_ld = d->cnt_ld;
    p.xyz_out = d->xyz_out; p.xyz_ld = d->xyz_ld;
    p.u2 = d->w2_unscale > 0.f ? d->w2_unscale : 1.f; p.u3 = d->w3_unscale > 0.f ? d->w3_unscale : 1.f;
    p.nblk = ceil_div(d->B * d->S, SAB_WAVES);
    if (d->feat) {
        EV2H_CHECK_ARG(d->W1f && d->b1 && d->nfeat >= 0 && d->nfeat <= 5 &&
                       d->ldf >= 8 && (d->ldf % 4) == 0 && d->ldw1f >= d->nfeat);
        p.feat = d->feat; p.ldf = d->ldf; p.W1f = d->W1f; p.ldw1f = d->ldw1f; p.b1 = d->b1; p.nfeat = d->nfeat;
        p.u1f = d->w1f_unscale > 0.f ? d->w1f_unscale : 1.f;
        p.u1x = d->w1x_unscale > 0.f ? d->w1x_unscale : 1.f;
        if ((d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_F16) && d->feat_amax) {
            EV2H_CHECK_ARG(d->dmax > 0.f && d->w1f_norm >= 0.f && d->b1_max >= 0.f && d->w1x_norm >= 0.f && d->w2_norm >= 0.f && d->b2_max >= 0.f);
            p.feat_amax = d->feat_amax; p.w1f_norm = d->w1f_norm; p.b1_max = d->b1_max;
            p.w1x_norm = d->w1x_norm; p.dmax = d->dmax; p.w2_norm = d->w2_norm; p.b2_max = d->b2_max;
        }
    } else {
        EV2H_CHECK_ARG(d->P1 != nullptr);
    }
    if (d->precision == EV2H_PREC_BF16) EV2H_CHECK_ARG(p.u2 == 1.f);      // (the BF16 layer-2 epilogue applies no factor)
    if (d->precision == EV2H_PREC_F16X2 || d->precision == EV2H_PREC_F16) {
        p.out_amax = d->out_amax;
        if (d->p1_scale) {
            EV2H_CHECK_ARG(d->p1_amax && d->dmax > 0.f && d->w1x_norm >= 0.f && d->w2_norm >= 0.f && d->b2_max >= 0.f);
            p.p1_scale = d->p1_scale; p.p1_amax = d->p1_amax;
            p.w1x_norm = d->w1x_norm; p.dmax = d->dmax; p.w2_norm = d->w2_norm; p.b2_max = d->b2_max;
        }
    }
    p.a1f = p.a1x = p.a1b = 1.f;
    if (d->precision == EV2H_PREC_F16 && (p.feat_amax || p.p1_scale)) {
        // F16 layer 1 (L1M): A = W / a with a = 2^(floor(log2 |W|_1) - 8) per k-slot group, B = s1 a x.  s1 keeps s1 (|W1f|_1 max|f| +
        // max|b1| + |W1x|_1 dmax) below 2^15 (the kernel's bound / the table's storage scale), so each group's B slots stay below
        // 2^15 a / |W|_1 <= 2^7 and each A entry below |W|_1 / a < 2^9: nothing overflows, whatever the checkpoint and the input
        auto pow2_below = [](float norm) { int e = 0; if (!(norm > 0.f) || !std::isfinite(norm)) return 1.f; (void)std::frexp(norm, &e); return std::ldexp(1.f, e - 1 - 8); };
        p.a1x = pow2_below(d->w1x_norm);
        if (d->feat) { p.a1f = pow2_below(d->w1f_norm); p.a1b = pow2_below(d->b1_max); }
    }
    hipStream_t st = (hipStream_t)stream;
    if (d->precision == EV2H_PREC_BF16X3) return dispatch_sab<3>(p, d->C1, d->C2, d->C3, st);
    if (d->precision == EV2H_PREC_F16X2) return dispatch_sab<2>(p, d->C1, d->C2, d->C3, st);
    if (d->precision == EV2H_PREC_BF16) return dispatch_sab<1>(p, d->C1, d->C2, d->C3, st);
    if (d->precision == EV2H_PREC_F16) return dispatch_sab<4>(p, d->C1, d->C2, d->C3, st);
    ev2h_set_error("ev2h_sa_mlp_max: unknown precision %d", d->precision);
    return EV2H_ERR_ARG;
}
