/* ev2hands_hip.h -- C ABI of libev2hands_hip.so: the MI355X (gfx950) implementation of the
 * Ev2Hands per-frame inference hot path (TEHNet.forward + MANO layer, left and right hand).
 *
 * The reference has no native code and no FFI: its hot path is PyTorch eager ops called from
 * /root/reference/src/Ev2Hands/model/model.py:61-64 (TEHNetWrapper.__call__ -> TEHNet.forward,
 * model/TEHNet.py:168-197).  This header is therefore the boundary WE define underneath that
 * Python interface (last-but-one row); each entry point names the reference
 * lines whose arithmetic it replaces.  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers (fp32 / int32 / int64), explicit sizes, no torch types;
 *  - the caller allocates and owns every input, output and workspace buffer;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *  - returns 0 on success, non-zero otherwise (EV2H_ERR_*); ev2h_last_error() gives the text;
 *    nothing throws across the ABI;
 *  - point-major layout inside the library: activations are [rows = points][channels] with the
 *    channel axis contiguous; only the boundary tensors keep the reference's [B, C, N] layout.
 */
#ifndef EV2HANDS_HIP_H
#define EV2HANDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EV2H_ABI_VERSION 8

typedef void* ev2h_stream_t; /* hipStream_t */

/* Arithmetic of the MFMA contractions (selections, biases, ReLU, max, MANO are always fp32):
 *  F32     v_mfma_f32_32x32x2_f32, exact fp32 products and accumulation;
 *  BF16X3  each fp32 operand split exactly into 3 bf16 planes, 6 plane products on
 *          v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-class accuracy (dropped terms O(2^-24));
 *  F16X2   each fp32 operand split into 2 fp16 planes (11 + 11 mantissa bits), 3 plane products on
 *          v_mfma_f32_32x32x16_f16 with fp32 accumulation: dropped terms O(2^-22).  ACCURACY CONTRACT (fp16 has 5 exponent bits):
 *          every operand tensor is scaled by ONE exact power of two per event window (activations, "Range records" below) or per
 *          matrix (weights, w_unscale), so its maximum sits in [2^14, 2^15) and nothing can overflow; a value keeps its full
 *          22 bits only down to 2^-17 of that maximum and an absolute error <= 2^-39 of the maximum below.  The mode is
 *          therefore fp32-class (<= 1e-5 relative on every output) only if no operand that matters lies more than 2^17 below
 *          its window's / matrix's maximum.  A checkpoint may distribute magnitude between a hidden channel and the weights
 *          that read it arbitrarily (BatchNorm scales): ev2h_pack_weights removes that freedom with an exact power-of-two
 *          per-channel equalisation (EV2H_PACK_EQUALIZE).  Measured WITHOUT it on otherwise identical networks: 6.6e-4 relative
 *          error with hidden channels 2^16 apart, 0.23 with channels 2^24 apart; with it <= 1.3e-6 up to 2^32 apart
 *          (tests/test_gpu_range.py, profiles/r3_spread_report.txt).  ev2h_forward refuses F16X2 weights that are not marked
 *          EV2H_W_EQUALIZED unless the caller opts out with EV2H_W_UNEQUALIZED_OK.  What no packer can see is a spread BETWEEN
 *          POINTS or between channels that only the data produces: ev2h_range_report counts, per tensor and window, the values
 *          that sit below 2^-17 of the maximum, and BF16X3 / F32 have no such limit;
 *  BF16    operands rounded to bf16 (RNE), fp32 accumulation (BASELINE.json config 3's arithmetic as named; 8 mantissa bits).  NOT usable
 *          on a trained checkpoint: 12-300 mm MPJPE against the exact-fp32 mode (profiles/r5_trained_precision_report.txt);
 *  F16     [ABI 8] the REDUCED-PRECISION mode to use (config 3): every operand rounded to ONE fp16 plane (RNE, 11 mantissa bits: 8 x
 *          finer than bf16) under F16X2's range machinery -- the same per-window / per-matrix powers of two from the same range
 *          records, so nothing overflows and a value keeps its 11 bits down to 2^-28 of its window's maximum; ONE product per
 *          multiply-add on v_mfma_f32_32x32x16_f16 (bf16's MFMA count and tile-image geometry); layer 1 of the set abstractions
 *          that read raw feature rows keeps F16X2's three plane products (coordinate differences: 5 % of a strip's MFMAs).
 *          Same equalisation contract as F16X2 (EV2H_W_EQUALIZED).  Measured on the optimiser-made checkpoints against the
 *          exact-fp32 mode: profiles/r6_trained_precision_report.txt;
 * A few small layers run as exact fp32 fma chains in EVERY mode, because the matrix pipe has nothing to gain there: the one-row-per-
 * window layers (ev2h_gemm_desc.skinny), the folded attention product (ev2h_attn_sim_folded) and -- where they are still computed
 * (F32; EV2H_L1_TABLE=1) -- the K = 8 layer-1 tables of the raw cloud.  In the three plane modes ev2h_forward runs layer 1 of the set
 * abstractions that read raw feature rows on the matrix pipe (ev2h_sa_desc.feat), without a table. */
#define EV2H_PREC_F32 0
#define EV2H_PREC_BF16 1
#define EV2H_PREC_F16X2 2
#define EV2H_PREC_BF16X3 3
#define EV2H_PREC_F16 4      /* [ABI 8] also the `planes` code of this mode in ev2h_tile_geometry / ev2h_pack_*_image (ONE plane is stored) */

/* Range records (EV2H_PREC_F16X2, EV2H_PREC_F16).  fp16 planes overflow at 65504, so every tensor that a contraction reads in F16X2 mode has a
 * "range record": uint32 [groups], the IEEE-754 bit pattern of max|value| over each group of rows (a group = the rows of one
 * event window), maintained with integer atomicMax by whichever kernels write the tensor (y_amax / out_amax / amax arguments;
 * the caller zeroes the records first).  A consumer scales each group by an exact power of two derived from the record
 * (x_amax / p1_amax arguments) before splitting, and multiplies the accumulated products back, so no plane can overflow
 * whatever the checkpoint or input magnitudes are, and a window's results never depend on the other windows of a batch.
 * NULL records switch the scaling off (operands are then split as they are and must stay below 65504). */

/* ---- library ------------------------------------------------------------------------------- */
int ev2h_abi_version(void);
/* [ABI 8] sha256 (hex) of the sources this binary was built from -- every file of csrc/ and every header of include/, names and contents, as
 * ev2hands_amd/build.py: source_hash() computes it -- and the extra compiler definitions of the build (EV2H_BUILD_DEFS; "" normally).
 * A binding that ships next to the sources compares the hash when it loads the library (ev2hands_amd/_lib.py does and refuses a
 * mismatch): the binary is git-ignored but copied between machines, so a stale one is otherwise silent. */
const char* ev2h_source_hash(void);
const char* ev2h_build_defs(void);
const char* ev2h_last_error(void);
/* One-time setup per device and host thread (raises dynamic-LDS limits of the big-tile kernels; creates the side stream
 * ev2h_forward forks onto).  Idempotent; ev2h_forward calls it itself.  Call it EARLY in a process that will create many HIP
 * streams (torch's stream pool, RCCL): HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and streams
 * that share one execute in order -- a side stream created late can land on the caller's queue, which silently removes the
 * two-stream overlaps (~5 % of a B = 256 step).  (Raising GPU_MAX_HW_QUEUES is no substitute: 8 queues cured the late-creation case
 * but cost 10 % when the side stream was created first.) */
int ev2h_init(void);
/* Per host thread: enabled = 0 makes ev2h_forward run every kernel on the caller's stream (no fork onto the library's side stream);
 * 1 (default) restores the two-stream schedule.  Results are bit-identical; returns the previous setting.  A host uses it to MEASURE
 * what the overlaps buy in its own process (bench.py's `two_stream_gain`): a gain of ~1.00 where ~1.05 is expected means the side
 * stream shares a hardware queue with the caller's stream (see ev2h_init). */
int ev2h_set_side_stream(int enabled);
/* Does the library's side stream really run CONCURRENTLY with `stream` in this process?  (HIP multiplexes streams onto a few
 * hardware queues; two streams that share one run in order, without any error -- see ev2h_init.)  Launches one short spin kernel
 * (~spin_us microseconds of s_sleep, no memory traffic) on `stream` alone, then one on each of the two streams at once, timed by
 * events; *ratio = (time of the pair) / (time of one): ~1.0 = concurrent, ~2.0 = the two streams are serialised.  Synchronises
 * `stream` (a start-up probe, ~0.2 ms); not capturable.  Returns EV2H_ERR_ARG when the side stream is switched off
 * (EV2H_TWO_STREAMS=0 / creation failed): then there is nothing to probe and *ratio is set to 0.  [ABI 7] */
int ev2h_side_stream_probe(ev2h_stream_t stream, int spin_us, float* ratio);
/* [ABI 8] Do two streams of this process run concurrently on the current device?  One spin kernel on `a` alone, then one on each of
 * `a` and `b` at once (b forked from a by an event, as ev2h_forward forks): *ratio = pair / one, ~1.0-1.3 concurrent, ~2 = the two
 * share a hardware queue and run in order.  Synchronises `a`; not capturable. */
int ev2h_streams_concurrent(ev2h_stream_t a, ev2h_stream_t b, int spin_us, float* ratio);
/* [ABI 8] Bind a side stream to `stream` that the device REALLY runs beside it (optional; once per caller stream, before or between
 * forwards -- ev2hands_amd calls it ahead of every forward, after the first time it returns at once).  The hardware queue a HIP
 * stream lands on depends on everything the process created before it (a framework's stream pool, RCCL, other libraries), so no
 * creation order is right for every host -- least of all with forwards in flight on several caller streams, each with a side
 * stream of its own.  This call MEASURES: it probes the slot's side stream against `stream` and against every stream the calling
 * thread has bound before (ev2h_streams_concurrent), and while a pair is serialised creates another candidate (at most 8) and keeps
 * the best; rejected candidates are destroyed.  Synchronises `stream` (~1-5 ms, once); does nothing while `stream` is capturing,
 * in single-stream mode, or while all side-stream slots (4 per host thread and device) serve other streams that are in use -- such a
 * stream shares the first slot's side stream (correct, only serialised); a slot that has been idle for eight forwards is recycled.
 * info (optional, int[3]): candidates tried, 1000 x ratio of the chosen pair, number of earlier-bound streams it still shares a queue with. */
int ev2h_bind_stream(ev2h_stream_t stream, int* info);
/* [ABI 8] What shader clock is the chip running at RIGHT NOW?  Enqueues a one-wave kernel on `stream` that sleeps for ~spin_us
 * microseconds and writes out_dev[0] = elapsed shader-clock cycles (s_memtime), out_dev[1] = elapsed ticks of the constant 100 MHz
 * reference counter (s_memrealtime): clock in MHz = 100 * out_dev[0] / out_dev[1].  The matrix-pipe kernels are power-limited on
 * real data (2.39 GHz idle boost -> ~1.65 GHz under a dense MFMA stream, profiles/r*_mfma_ceiling.txt), so a throughput figure taken
 * over a fraction of a second can ride on a clock that a production run does not keep: a host launches this on a second stream
 * while its workload runs and reports the clock next to the rate (bench.py `value_sustained`).  Asynchronous; capturable. */
int ev2h_shader_clock_probe(ev2h_stream_t stream, int spin_us, unsigned long long* out_dev);
/* sizeof() of the structs below, for binding-side layout checks: [gemm, sa, sa_module, weights, mano, outputs, fp, tensor_desc]. */
void ev2h_struct_sizes(size_t out[8]);

/* ---- point-set operators --------------------------------------------------------------------- */
/* [B,C,N] channel-major input -> pts4 [B][N][4] = (x, y, z, (x*x+y*y)+z*z) and feat8 [B][N][8]
 * (the C input channels, zero padded).  mhlnes != 0 reproduces TEHNet.py:176-177, including the
 * in-place overwrite of input channel 2. */
int ev2h_prep_points(float* xyz_cm, int B, int C, int N, int mhlnes, float* pts4, float* feat8, uint32_t* feat_amax,
                     ev2h_stream_t stream);       /* feat_amax: optional range record [B] of the input channels */

/* farthest_point_sample, model/pointnet2_utils.py:63-84.  init [B] int64 start indices (the
 * reference draws them with torch.randint on the host, :75).  Writes idx [B][S] and the selected
 * points ctr4 [B][S][4]. */
int ev2h_fps(const float* pts4, int B, int N, int S, const int64_t* init, int32_t* idx, float* ctr4, ev2h_stream_t stream);
/* Up to 3 independent samplings of the same clouds in one launch (host arrays of length njobs). */
int ev2h_fps_multi(const float* pts4, int B, int N, int njobs, const int* S, const int64_t* const* init,
                   int32_t* const* idx, float* const* ctr4, ev2h_stream_t stream);

/* query_ball_point, model/pointnet2_utils.py:87-107, for up to 3 radii of one centroid set in one
 * pass.  radius / nsample / gidx are HOST arrays of length nrad; gidx[i] is a device buffer
 * [B][S][nsample[i]] int32; cnt (optional, device) [B][S][nrad] receives min(#in-radius, nsample).
 * radius is DOUBLE: the reference compares fp32 distances with `radius ** 2` evaluated in Python double precision and then
 * rounded to fp32 (:102); squaring an fp32-rounded radius gives a threshold one ulp higher for 0.1 and 0.2, which flips
 * points that lie exactly on the radius. */
int ev2h_ball_query(const float* pts4, const float* ctr4, int B, int N, int S, int nrad, const double* radius,
                    const int* nsample, int32_t* const* gidx, int32_t* cnt, ev2h_stream_t stream);

/* 3-NN inverse-distance interpolation, model/pointnet2_utils.py:296-303.  pts1 [B][N1][4] are the
 * query points, pts2 [B][N2][4] the known points with features feat2 [B][N2][ldf2] (first D used).
 * out [B][N1][ldo] (first D written; may be NULL), nn_idx / nn_w [B][N1][3] optional. */
int ev2h_three_nn_interp(const float* pts1_4, const float* pts2_4, int B, int N1, int N2, const float* feat2, int ldf2,
                         int D, float* out, int ldo, int32_t* nn_idx, float* nn_w, uint32_t* out_amax, ev2h_stream_t stream);
                         /* out_amax: optional range record [B] of the interpolated rows */

/* ---- dense layers ------------------------------------------------------------------------------ */
/* Y[m][n] = post( relu?( sum_{t<taps} sum_{k<K} X[m + t - (taps==3)][k] * W[n][t*K + k] + bias[n] ) )
 * Replaces every Conv1d/Conv2d(1x1)/Linear (+ eval BatchNorm, folded by the caller or applied as
 * post_scale/post_shift) of model/TEHNet.py:49-55,129-141,150-166 and model/pointnet2_utils.py:195-199,312-314. */
typedef struct ev2h_gemm_desc {
    const float* X; int ldx;     /* [M][ldx], K (x taps) used                                         */
    const float* W; int ldw;     /* [N][ldw]                                                            */
    float* Y; int ldy;           /* [M][ldy]  (or [M / rowmax_rows][ldy] when rowmax_rows != 0)         */
    int M, N, K;                 /* K per tap, multiple of 4                                            */
    const float* bias;           /* [N] or, with bias_group_rows, [M / bias_group_rows][ldbias]; NULL ok */
    int bias_group_rows;         /* 0, or a multiple of 128                                             */
    int ldbias;
    int relu;
    const float* post_scale;     /* optional y = y * post_scale[n] + post_shift[n] AFTER the ReLU       */
    const float* post_shift;
    int taps;                    /* 1, or 3 = Conv1d(k=3, padding=1) along rows, zero padded per sequence */
    int rows_per_seq;            /* rows per window when taps == 3                                      */
    int rowmax_rows;             /* 0, or 128: write max over each 128-row group (group-all set abstraction) */
    int precision;               /* EV2H_PREC_*; all but F32 need K % 8 == 0 (operands are split on the fly), K % 16 == 0 with taps == 3 */
    const void* Ws;              /* optional, BF16 / BF16X3: bf16 plane images of W in (ws_tile_rows)-row x 32-k LDS
                                    tiles (ev2h_pack_gemm_image); NULL = split W on the fly  */
    int ws_tile_rows;            /* 128 (three 4-wave workgroups per CU) or 256 (one 8-wave workgroup)           */
    float w_unscale;             /* 16-bit precisions: W is used as W / w_unscale (Ws holds those planes) and the product is
                                    multiplied by w_unscale before the bias; a power of two chosen by the host so that the
                                    fp16 planes of small weights are not subnormal (ev2h_plane_unscale).  0 = 1.      */
    /* F16X2 activation range (ignored by the other precisions; see "Range records" below).  All optional. */
    const uint32_t* x_amax;      /* [ceil(M / x_group_rows)] range record of X: rows of group g are multiplied by the power of   */
    const uint32_t* x_amax2;     /* two that puts max(x_amax[g], x_amax2[g]) in [2^14, 2^15) before the fp16 split (x_amax2:     */
    int x_group_rows;            /* second source of a concatenated X, may be NULL); products are multiplied back.  taps == 3:   */
                                 /* x_group_rows must be a multiple of rows_per_seq.                                             */
    uint32_t* y_amax;            /* [..] range record of Y, updated with atomicMax (zero it before the first producer runs);      */
    int y_group_rows;            /* groups of y_group_rows OUTPUT rows (with rowmax_rows: rows of the reduced output)             */
    float* y_scale;              /* out [ceil(M / y_group_rows)], needs x_amax and y_group_rows == x_group_rows: Y is            */
    float y_bound_w, y_bound_b;  /* stored times the power of two s[g] with (y_bound_w * max|X_g| + y_bound_b) * s[g] in          */
                                 /* [2^14, 2^15) -- y_bound_w >= max row L1 norm of W makes that a bound of |Y| (+ whatever the   */
                                 /* caller adds to y_bound_b); used for the layer-1 tables of ev2h_sa_mlp_max                     */
    int skinny;                  /* 1: the layer has ONE ROW PER WINDOW (M = B: fp3's broadcast half, the Linear layers of the MANO heads,
                                    TEHNet.py:49-55, pointnet2_utils.py:293-294): a kernel that parallelises over K instead of over rows, exact
                                    fp32 fma chains in every precision (x_amax is not needed, y_amax is still maintained).  Needs taps == 1,
                                    no rowmax / per-group bias / y_scale.  The caller sets it by layer, never by batch size.     */
} ev2h_gemm_desc;
int ev2h_gemm(const ev2h_gemm_desc* d, ev2h_stream_t stream);

/* [B*N][4] point-major logits -> class_logits [B,4,N] (TEHNet.py:188,197).  cm_window_stride: floats between consecutive windows of
 * logits_cm (0 = 4 * N, dense). */
int ev2h_transpose_logits(const float* logits_pm, int B, int N, float* logits_cm, size_t cm_window_stride, ev2h_stream_t stream);

/* ---- fused grouped set-abstraction MLP ----------------------------------------------------------- */
/* out[b][s][c] = max_k relu(W3' relu(W2' relu(P1[b][gidx[b][s][k]] + W1x (xyz[idx] - ctr[b][s])) + b2') + b3')
 * = model/pointnet2_utils.py:244-257 for one radius branch, with eval-BN folded into W/b and the
 * feature part of layer 1 pre-computed per point (P1 = W1f' f + b1').
 * Supported (C1,C2,C3): (32,32,64) (64,64,128) (64,96,128) (128,128,256) (128,196,256). */
typedef struct ev2h_sa_desc {
    const float* P1; int ldp;    /* [B][Npts][ldp], C1 columns used                                     */
    const float* pts4;           /* [B][Npts][4]                                                        */
    const float* ctr4;           /* [B][S][4]                                                           */
    const int32_t* gidx;         /* [B][S][K]                                                           */
    const float* W1x;            /* [C1][4]  folded weights of the 3 relative-xyz inputs (4th = 0)      */
    const float* W2;             /* [roundup(C2,32)][C1], zero-padded rows                              */
    const float* b2;             /* [roundup(C2,32)]                                                    */
    const float* W3;             /* [C3][roundup(C2,8)], zero-padded columns                            */
    const float* b3;             /* [C3]                                                                */
    float* out; int ldo;         /* [B][S][ldo], C3 columns written                                     */
    int B, Npts, S, K;           /* K multiple of 32                                                    */
    int C1, C2, C3;
    int precision;               /* EV2H_PREC_*                                                         */
    const void* W2s;             /* bf16 tile images of W2 / W3 for the 16-bit modes (ev2h_pack_sa_images), */
    const void* W3s;             /* NULL for F32                                                        */
    const int32_t* cnt;          /* optional: cnt[(b*S + s) * cnt_ld] = number of distinct neighbours of the group   */
    float w2_unscale, w3_unscale;/* power-of-two factors the W2s / W3s planes were divided by (0 = 1), see ev2h_gemm_desc   */
    int cnt_ld;                  /* (ev2h_ball_query's count output); slots >= cnt repeat slot 0, so whole 32-slot
                                    strips of padding are skipped by the 16-bit kernels -- the max is unchanged.  NULL (or
                                    EV2H_PREC_F32) = process all K slots                                       */
    /* F16X2 activation range (optional; see "Range records").  The layer-1 table arrives scaled: P1 holds s[b] * (W1f' f + b1')
     * with the power of two p1_scale[b] = s[b] chosen by the table's producer (ev2h_gemm y_scale) such that
     * s[b] * (max|P1_b| + w1x_norm * dmax) < 2^15, and p1_amax[b] = max|stored P1_b|.  The hidden activations of layers 1 and 2
     * are then kept inside the fp16 range by exact per-window powers of two derived from these bounds.
     * EV2H_PREC_F16 [r6]: ONE power of two per window serves the whole chain (layer 2's accumulators are converted without a factor
     * of their own), so the producer's s[b] must ALSO satisfy (s[b] / w2_unscale) * (w2_norm * B1 + b2_max) < 2^15 with
     * B1 = max|P1_b| / s[b] + w1x_norm * dmax (ev2h_pack_sa_images chooses w2_unscale = 2^floor(log2 w2_norm) in this mode, so the two
     * conditions agree to a factor 2-4).  A window whose scale breaks this gets NaN outputs, not saturated ones. */
    const float* p1_scale;       /* [B]; NULL = P1 unscaled, no range handling                                          */
    const uint32_t* p1_amax;     /* [B]                                                                                 */
    float w1x_norm;              /* max_c (|W1x[c][0]| + |W1x[c][1]| + |W1x[c][2]|)                                     */
    float dmax;                  /* contract: every grouped neighbour lies within dmax of its centroid per coordinate
                                    (the ball-query radius); if it is broken, hidden values are saturated at 65504
                                    (F16X2) or overflow to inf and the outputs to NaN (F16: no clamp, one VALU op per pair)  */
    float w2_norm;               /* max row L1 norm of W2                                                               */
    float b2_max;                /* max |b2|                                                                            */
    uint32_t* out_amax;          /* [B] range record of `out` (atomicMax), optional                                     */
    /* plane modes (BF16, F16X2, BF16X3), optional: layer 1 straight from the raw feature rows, on the matrix pipe -- P1 is then not needed (may be NULL)
     * and no layer-1 table has to be computed or gathered: feat [B][Npts][ldf] (first nfeat <= 5 columns used, ldf >= 8: the forward's
     * feat8 / hf8 rows), W1f [C1][ldw1f] and b1 [C1] the folded layer-1 feature weights and bias of this branch.
     * BF16: inputs enter as two bf16 planes (16 bits), weights as bf16.  BF16X3: inputs and weights as their exact three bf16 planes,
     * the six plane products of the mode, no factors.  F16X2: every neighbour's feature values and relative xyz are
     * scaled by powers of two of their OWN (derived from that neighbour's maxima and from the two weight blocks' plane factors
     * w1f_unscale / w1x_unscale = ev2h_plane_unscale of W1f and of W1x), so a neighbour's values keep 22 bits relative to its own
     * largest term whatever the rest of the window holds; range handling (optional): feat_amax [B] = the range record of the feature rows,
     * w1f_norm / b1_max = max row L1 norm of W1f / max |b1| (with w1x_norm, dmax, w2_norm, b2_max above; p1_scale / p1_amax unused). */
    const float* feat; int ldf;
    const float* W1f; int ldw1f;
    const float* b1;
    int nfeat;
    const uint32_t* feat_amax;
    float w1f_norm, b1_max, w1f_unscale, w1x_unscale;
    /* optional (ABI 7): also write every group's centroid as (x, y, z, 0, 0, 0, 0, 0) to xyz_out[(b * S + s) * xyz_ld .. + 8) -- the
     * raw-xyz columns of the group-all layer that consumes `out` (model/pointnet2_utils.py:155 concatenates [xyz, features]; the
     * forward's buffers hold [features | xyz | pad]), which used to take a launch of their own per module.  xyz_ld % 4 == 0.     */
    float* xyz_out; int xyz_ld;
    /* optional [ABI 8], 16-bit plane precisions: this launch covers the centroids [s_off, s_off + S) of every window, while ctr4, gidx, cnt,
     * out and xyz_out are laid out for S_total centroids per window (S_total = 0: S, the whole set).  ev2h_forward draws enc.sa1's
     * 512 centroids in four chunks when the batch is too small to fill the chip, and runs the grouping + the fused MLP of the
     * centroids already drawn beside the rest of the sampling (512 dependent steps on a fraction of the CUs). */
    int S_total, s_off;
} ev2h_sa_desc;
int ev2h_sa_mlp_max(const ev2h_sa_desc* d, ev2h_stream_t stream);

/* ---- fused row chains: feature propagation and the segmentation head ---------------------------------------------------
 * (a) nn_idx != NULL -- PointNetFeaturePropagation with points1 = None (model/pointnet2_utils.py:296-316; fp1 of TEHNet.py:186):
 *   out[b][n] = relu(W3' relu(W2' relu(sum_j w[b][n][j] T[b][idx[b][n][j]]) + b2') + b3'),   T = W1' points2 + b1' per coarse point:
 *   the first Conv1d commutes with the 3-NN interpolation (a row's weights sum to 1 up to rounding, so carrying b1' inside the
 *   table changes a value by <= 2e-7 |b1'|), which makes layer 1 a table over the S coarse points instead of a GEMM over the N
 *   fine ones, and layers 2-3 run in registers like ev2h_sa_mlp_max -- the interpolated rows and both hidden layers never
 *   reach memory.  nn_idx / nn_w as written by ev2h_three_nn_interp.  (C1, C2, C3) = (128, 128, 256).
 * (b) nn_idx == NULL -- two dense layers on plain rows (the classifier, TEHNet.py:135-141,188: Conv1d -> ReLU -> BN -> Conv1d with the
 *   BN folded FORWARD into the second convolution by the host, exact for a k=1 convolution):
 *   out[b][n] = W3' relu(W2' T[b][n] + b2') + b3'  (+ ReLU unless no_relu_out),   T [B][N][ldt] = the input rows themselves;
 *   (C1, C2, C3) = (256, 256, 32) with out_cols <= C3 leading columns written (zero-padded W3' rows), optionally also channel-major.
 * 16-bit plane precisions only (EV2H_PREC_F32: ev2h_three_nn_interp + ev2h_gemm). */
typedef struct ev2h_fp_desc {
    const float* T; int ldt;      /* (a) [B][S][ldt] layer-1 table of the coarse points; (b) [B][N][ldt] input rows; C1 columns used */
    const int32_t* nn_idx;        /* [B][N][3], or NULL: form (b)                                                     */
    const float* nn_w;            /* [B][N][3]                                                                        */
    const float* b2;              /* [roundup(C2, 32)]                                                                */
    const float* b3;              /* [C3]                                                                             */
    const void* W2s;              /* tile images as in ev2h_sa_desc                                                   */
    const void* W3s;
    float w2_unscale, w3_unscale;
    float* out; int ldo;          /* [B][N][ldo], out_cols (0 = C3) columns written                                   */
    int B, N, S;                  /* S: coarse points per window, form (a)                                            */
    int C1, C2, C3;
    int precision;
    /* F16X2 range (optional, as in ev2h_sa_desc).  (a): the table arrives scaled by the power of two t_scale[b] (ev2h_gemm
     * y_scale) and t_amax is the record of the stored table; (b): t_scale = NULL, t_amax = the record of the input rows. */
    const float* t_scale;         /* [B]                                                                              */
    const uint32_t* t_amax;       /* [B]; NULL = no range handling                                                    */
    float w2_norm, b2_max;
    uint32_t* out_amax;           /* [B] range record of `out`, optional                                              */
    int out_cols;                 /* 0 = C3                                                                           */
    int no_relu_out;              /* 1: no ReLU after the last layer                                                  */
    float* out_cm;                /* optional second copy of the output, channel-major [B][out_cols][N]               */
    size_t out_cm_stride;         /* floats between consecutive windows of out_cm (0 = out_cols * N, dense)           */
} ev2h_fp_desc;
int ev2h_fp_mlp(const ev2h_fp_desc* d, ev2h_stream_t stream);

/* Geometry of the host-packed weight tile images (W2s / W3s of ev2h_sa_desc and ev2h_fp_desc, Ws of ev2h_gemm_desc) for a chain
 * (C1, C2, C3) and `planes` operand planes (1 BF16, 2 F16X2, 3 BF16X3; 4 = the F16 mode's code: one fp16 plane, BF16's geometry), straight from the kernels' compile-time configuration:
 * out = { T2 (32-row layer-2 tiles), C2P (layer-3 contraction length in the permuted order), RS2, RS3 (bytes per LDS row of a
 * W2 / W3 tile), TB2, TB3 (bytes per tile), GEMM RS (bytes per row of a dense W image tile), GEMM BK (k columns per tile),
 * LEFTOVER (0, or the 1..4 channels beyond the last full 32-channel tile whose plane products share MFMAs: the images then carry,
 * in the HIGH plane, the low plane of the leftover W2 rows in rows 8..11 of the last tile and [wh | wh | wl | 0] in the last 16
 * k-slots of every W3 row -- csrc/sa_mlp_bf16.hip SaBCfg::PACK4), W2PERM (1 = BF16: inside every 32-column chunk of a W2 image,
 * position 16h + 8m + e holds input channel 16m + 4h + (e & 3) + 8(e >> 2) -- the D-register order of the layer-1 MFMA; 0 = plain
 * channel order) }.
 * The library's own packer (ev2h_pack_weights, ev2h_pack_sa_images) asserts its layout against this on every pack, so the kernels
 * and the packer cannot drift apart silently; a caller that builds images itself can do the same.  Needs no GPU. */
int ev2h_tile_geometry(int C1, int C2, int C3, int planes, int out[10]);

/* ---- attention (model/TEHNet.py:13-27) ------------------------------------------------------------ */
/* sim[b][h][c][d] = softmax_c( 256^-0.5 * sum_n logits[b][n][c] * query_h[b][n][d] ); query of hand h
 * is query_pm + h*256 with row stride ldq.  logits_pm [B*N][4]; sim [B][2][4][256]. */
int ev2h_attn_sim(const float* logits_pm, const float* query_pm, int ldq, size_t query_hand_stride, int B, int N, float* sim,
                  ev2h_stream_t stream);
/* The same sim with the LAST query convolution folded behind the sum over the points.  query = Conv1d(k=3) -> ReLU -> BN ->
 * Conv1d(k=3) -> BN (TEHNet.py:150-166) enters the attention only through sum_n key[c][n] query[d][n] (TEHNet.py:20), and its last
 * Conv1d -> BN is affine in q1 (the first block's output): with W, b the BN-folded second convolution,
 *     sum_n key[c][n] query[d][n] = sum_t sum_i W[d][t][i] Z[c][t][i] + b[d] K[c],
 *     Z[c][t][i] = sum_n key[c][n] q1[i][n + t - 1] (zero padded),   K[c] = sum_n key[c][n]
 * -- one streaming pass over q1 and a [4 x 768] x [768 x 256] product per (window, hand) instead of an [N x 768] x [768 x 256]
 * convolution per hand; exact fp32 fma chains in a fixed order (no atomics).  A re-association of the reference's sum: results
 * agree to fp32 rounding, not bit for bit.
 * q1_pm [B*N][ldq]: hand h's 256 channels at column h*256.  w4t_* [768][256] = W transposed (row t*256 + i, column d), b4_* [256].
 * scratch: ev2h_attn_sim_folded_scratch(B, N) floats.
 * (ev2h_forward in BF16 / F16X2 on windows of a multiple of 128 points does not call this operator for the Z sums: the first query
 * convolution's GEMM forms them in its epilogue and q1 is never written -- csrc/gemm_bf16.hip: zsum_epilogue; EV2H_ATTN_UNFUSED_ZSUM=1
 * restores the operator form.  Same contract, different summation order: 5e-7 apart in F16X2.) */
size_t ev2h_attn_sim_folded_scratch(int B, int N);
int ev2h_attn_sim_folded(const float* logits_pm, const float* q1_pm, int ldq, int B, int N, const float* w4t_left,
                         const float* w4t_right, const float* b4_left, const float* b4_right, float* scratch, float* sim,
                         ev2h_stream_t stream);
/* hf8[h][b*N + n][0..3] = sum_d sim[b][h][c][d] * value[b*N+n][d]; columns 4..7 are written as 0.
 * value_unscale: optional [256], value channel d is used as value[..][d] * value_unscale[d] (exact when the factors are powers of
 * two: the host's channel equalisation stores fp1's output scaled per channel). */
int ev2h_attn_context(const float* sim, const float* value_pm, int ldv, int B, int N, float* hf8, uint32_t* hf_amax,
                      int amax_hand_stride, const float* value_unscale, ev2h_stream_t stream);
                      /* hf_amax: optional range records, hand h at hf_amax[h * amax_hand_stride + b] */

/* ---- MANO layer (manopth ManoLayer.forward behind model/utils.py:25-31) ----------------------------- */
typedef struct ev2h_mano_consts {
    const float* hands_mean;     /* [45]                                                               */
    const float* comps;          /* [ncomps][45]                                                       */
    const float* blend_T;        /* [145][2336]: rows 0..9 shapedirs, 10..144 posedirs; column v*3+c   */
    const float* v_template;     /* [2334]                                                             */
    const float* J_template;     /* [48]      J_regressor @ v_template                                 */
    const float* J_shape;        /* [10][48]  J_regressor @ shapedirs                                  */
    const float* weights;        /* [778][16] skinning weights                                         */
    int32_t tips[5];
    int32_t ncomps;
} ev2h_mano_consts;
/* params [B][ldp] = global_orient(3) | hand_pose(ncomps) | betas(10) | transl(3)  (TEHNet.py:87-90).
 * verts [B][778][3], joints [B][21][3] in metres (mm scaling and /1000 of utils.py:28-29 included).
 * verts_stride / joints_stride: floats between consecutive windows (0 = 2334 / 63, dense). */
int ev2h_mano(const ev2h_mano_consts* c, const float* params, int ldp, int B, float* verts, size_t verts_stride, float* joints,
              size_t joints_stride, ev2h_stream_t stream);
/* Parity access: rot [B][16][9] = the rotation matrices ev2h_mano derives from params (global orientation + 15 articulated joints;
 * axis-angle -> matrix by the formula of losses.py:14-51: quaternion route, theta + 1e-8 inside the norm).  Only hands_mean, comps
 * and ncomps of `c` are used. */
int ev2h_mano_rotations(const ev2h_mano_consts* c, const float* params, int ldp, int B, float* rot, ev2h_stream_t stream);

/* ---- event window -> [5, N] tensor (next row 8f-1; dataset/evaluation_stream.py:187-225, ev2hands_r.py:108-159, erpc.py:169-249) ---- */
/* Per-pixel accumulation + np.nonzero-order compaction of B ragged windows.  events: device float64 rows of ev_stride (>= 4)
 * columns (x, y, t, polarity, ...) in stream order, exactly the arrays the reference builds; offsets: device [B+1] row offsets
 * (<= 32768 events per window).  uniq [B][cap][8] float32 records (x, y, t_avg, pos_cnt, neg_cnt, 0, 0, 0) of the pixels hit, in
 * row-major pixel order; uniq_count [B] (-1 if a window is too large).  Bit-identical to np.add.at / np.nonzero.
 * raw_time = 0: evaluation builders (t minus the window's first timestamp, evaluation_stream.py:187);
 * raw_time = 1: Ev2Hands-S builder (timestamps as they are, mean times 1e-6, erpc.py:178-191). */
int ev2h_event_window_build(const double* events, int ev_stride, const int32_t* offsets, int B, int width, int height, int cap,
                            int raw_time, int32_t* uniq_count, float* uniq, ev2h_stream_t stream);
/* Ev2Hands-S only (erpc.py:207-211): unique pixels re-ordered by mean time (np.argsort; exactly equal times keep pixel order --
 * numpy leaves their order undefined), first time subtracted.  uniq_out must differ from uniq_in; cap <= 16384.  labels_out
 * [B][cap] (optional) = events[offsets[b] + perm[j]][label_col]: the per-event label column indexed with the per-pixel sort
 * positions, as erpc.py:209 does. */
int ev2h_event_window_timesort(const float* uniq_in, const int32_t* uniq_count, int cap, const double* events, int ev_stride,
                               int label_col, const int32_t* offsets, int B, float* uniq_out, int32_t* labels_out,
                               ev2h_stream_t stream);
/* Resampling + pc_normalize: sample_idx [B][N] int32 (the reference draws them with np.random.choice on the host; for erpc.py's
 * sampling=False branch pass arange(M) followed by the N - M drawn indices) -> out_cm [B][5][N] float32 = (x, y, t, pos_cnt,
 * neg_cnt), the hot path's input.  uniq_labels [B][cap] / out_labels [B][N] int64 optional (Ev2Hands-S 'class_logits' target). */
int ev2h_event_window_sample(const float* uniq, const int32_t* uniq_count, int cap, const int32_t* sample_idx, int B, int N,
                             int width, int height, float* out_cm, const int32_t* uniq_labels, int64_t* out_labels,
                             ev2h_stream_t stream);

/* ---- per-frame joint metrics (next row 8f-3; evaluate.py:185-234, evaluate_ev2hands_r.py:35-89) ------------------------ */
/* j3d_left / j3d_right [B][21][3] float32 metres (the forward's outputs); j3d_gts [B][G][2][21][3] float64 metres (G ground-truth
 * candidates per frame).  For the candidate with the best rounded right-root-relative AUC (first on ties): pck [B][3][num_steps+1]
 * (absolute, relative, right-root-relative), auc [B][3] (trapezoid / n, NOT yet rounded), mpjpe [B] (mm), root_distance [B] (mm),
 * best [B]. */
int ev2h_joint_metrics(const float* j3d_left, const float* j3d_right, const double* j3d_gts, int B, int G, int num_steps,
                       double dist_max_mm, float* pck, double* auc, double* mpjpe, double* root_distance, int32_t* best,
                       ev2h_stream_t stream);

/* ---- two-hand mesh self-collision (next row 8f-4; evaluate_ev2hands_r.py:128-160, utils/__init__.py:106-124) ---------------- */
/* verts_left / verts_right [B][nv][3] float32 metres (the forward's vertices), faces [nf][3] int32 (nv <= 778, nf <= 1538).
 * The meshes are concatenated as the reference does (left faces, then right faces + nv), vertices scaled by `scale` (1000:
 * mm) in float32 and tested in float64.  counts [B] = number of unordered triangle pairs that share no vertex index and
 * intersect (separating-axis test, touching counts); pairs [B][max_pairs][2] (optional) = the first max_pairs of them in
 * lexicographic (i < j) order.  The reference obtains its pairs from the un-vendored torch-mesh-isect BVH with a
 * per-triangle candidate cap; this is the uncapped quantity (parity unpinned, oracle/collision_oracle.py). */
int ev2h_mesh_collisions(const float* verts_left, const float* verts_right, const int32_t* faces_left, const int32_t* faces_right,
                         int B, int nv, int nf, float scale, int max_pairs, int32_t* pairs, int32_t* counts, int max_per_triangle,
                         ev2h_stream_t stream);
/* The same search with a caller-owned scratch buffer (device, ev2h_mesh_collisions_scratch_bytes(B, nf) bytes): with at most 128
 * windows the row blocks of a window are then split over TWO workgroups (one 1024-thread workgroup with 142 KB of LDS per window
 * leaves half of the 256 CUs idle at BASELINE config 5's 128 windows per GPU); counts and pair lists are identical entry for entry. */
size_t ev2h_mesh_collisions_scratch_bytes(int B, int nf);
int ev2h_mesh_collisions_ws(const float* verts_left, const float* verts_right, const int32_t* faces_left, const int32_t* faces_right,
                            int B, int nv, int nf, float scale, int max_pairs, int32_t* pairs, int32_t* counts, int max_per_triangle,
                            void* scratch, size_t scratch_bytes, ev2h_stream_t stream);
/* max_per_triangle > 0: at most that many pairs (i, j > i) are counted and listed per triangle i, the first ones in j order -- the
 * role of the reference BVH's `max_collisions` (8 in evaluate_ev2hands_r.py:131, 16 in losses.py:62).  Which pairs the reference's
 * tree keeps beyond its cap depends on its traversal order (not reproducible); the two agree whenever no triangle exceeds the cap. */

/* Penetration penalty of the listed pairs (losses.py:60-102 CollisionLoss: torch-mesh-isect DistanceFieldPenetrationLoss with
 * sigma = 0.5, point2plane = False, penalize_outside = False; restated from Tzionas et al. IJCV 2016 eq. 11-14, parity unpinned).
 * pairs / counts / max_pairs as written by ev2h_mesh_collisions for the same meshes and scale (1.0: the loss works in metres).
 * loss [B] float64 = sum over the window's pairs of both triangles' conic distance-field terms; the caller applies the
 * reference's reduction (mean over the windows with a non-zero loss, times collision_weight = 100). */
int ev2h_collision_penalty(const float* verts_left, const float* verts_right, const int32_t* faces_left, const int32_t* faces_right,
                           int B, int nv, int nf, float scale, double sigma, const int32_t* pairs, const int32_t* counts, int max_pairs,
                           double* loss, ev2h_stream_t stream);

/* ---- whole path -------------------------------------------------------------------------------------- */
typedef struct ev2h_sa_branch {
    const float* W1x; const float* W2; const float* b2; const float* W3; const float* b3;
    int C1, C2, C3, K;
    double radius;                      /* double: see ev2h_ball_query */
    const void* W2s; const void* W3s;   /* 16-bit tile images (NULL unless precision != F32) */
    float w2_unscale, w3_unscale;       /* see ev2h_sa_desc */
    float w1x_norm, w2_norm, b2_max;    /* F16X2 range bounds as in ev2h_sa_desc; filled by ev2h_pack_weights */
    float w1f_unscale, w1x_unscale;     /* F16X2: power-of-two plane factors of this branch's W1f and W1x (ev2h_sa_desc.w1f_unscale / w1x_unscale) */
    float w3_norm, b3_max;              /* [ABI 8] max row L1 norm of W3, max |b3|: the bound of the chain's OUTPUT (F16: fp1's output rows are
                                           stored as fp16 times a power of two chosen from it)                                       */
} ev2h_sa_branch;

typedef struct ev2h_sa_module {   /* one PointNetSetAbstractionMsg (pointnet2_utils.py:205-262)       */
    const float* W1f;             /* [sum C1][kf]  layer-1 feature weights of all branches, BN folded  */
    const float* b1;              /* [sum C1]                                                          */
    int kf;                       /* padded feature width of the input table                           */
    int npoint;
    int nbranch;
    ev2h_sa_branch br[3];
    float w1f_unscale;            /* power-of-two plane factor of W1f (ev2h_gemm_desc.w_unscale)       */
    const void* W1fs;             /* optional 128-row plane images of W1f (kf >= 32: enc.sa2), else NULL */
    float w1f_norm, b1_max;       /* F16X2 range bounds of the table: max row L1 norm of W1f, max |b1| */
} ev2h_sa_module;

typedef struct ev2h_dense {       /* one folded Conv/Linear: W [O][ldw], b [O], optional post affine    */
    const float* W; const float* b; const float* post_scale; const float* post_shift;
    int O, K, ldw;
    const void* Ws;               /* bf16 plane images of W (NULL: split on the fly / F32)              */
    int ws_tile_rows;             /* rows per image tile: 128 or 256                                    */
    float w_unscale;              /* see ev2h_gemm_desc                                                  */
} ev2h_dense;

typedef struct ev2h_weights {
    ev2h_sa_module sa1, sa2, mano_sa1[2];       /* [0] = left, [1] = right                             */
    ev2h_dense sa3[3];                           /* 520(=512 feat|xyz|pad) -> 256 -> 512 -> 1024       */
    ev2h_dense fp3_skip, fp3_bcast, fp3_1;       /* 1536 split into 512 (skip) + 1024 (broadcast l3)   */
    ev2h_dense fp2[2];
    ev2h_dense fp1[3];
    ev2h_sa_module fp1m;                         /* fp1 once more in the form ev2h_fp_mlp takes (16-bit precisions): W1f/b1 = table
                                                    layer, br[0] = layers 2-3 (W1x unused); kf = 128, nbranch = 1               */
    ev2h_dense cls0, cls4;
    ev2h_sa_branch clsm;                         /* the classifier once more as an ev2h_fp_mlp chain, form (b) (16-bit precisions):
                                                    W2s = cls0, W3s = cls4 with the BN between them folded in, C3 = 32 (4 used)  */
    ev2h_dense qconv0;                           /* both hands, O = 512, 3 taps                        */
    ev2h_dense qconv4[2];
    const float* qconv4T[2];                     /* the same folded weights transposed, [768][256] (ev2h_attn_sim_folded)      */
    ev2h_dense mano_sa2[2][2];
    ev2h_dense head0[2], head4[2];
    int precision;                               /* EV2H_PREC_* used by the MFMA kernels               */
    const float* l0_unscale;                     /* [256] or NULL: ev2h_attn_context's value_unscale -- the per-channel powers of two
                                                    the channel equalisation (ev2h_pack_weights) multiplied fp1's
                                                    output by; every other consumer of that tensor has them folded into its columns */
    int flags;                                   /* EV2H_W_*                                            */
    int f16_families;                            /* [ABI 8] precision == EV2H_PREC_F16 only: EV2H_FAM_* mask of the kernel families
                                                    that run on ONE fp16 plane; the others run F16X2 (their images are packed
                                                    accordingly; both read the same range records).  ev2h_pack_weights sets it.      */
} ev2h_weights;
/* kernel families of the F16 mode (ev2h_weights.f16_families; EV2H_PACK_F16_FAMILIES) */
#define EV2H_FAM_SA 1             /* the fused set abstractions (enc.sa1, enc.sa2 + its layer-1 table GEMM, both regressors' sa1): 66 % of the MACs */
#define EV2H_FAM_ROWS 2           /* the row chains: fp1 (+ its table GEMM) and the segmentation head                               */
#define EV2H_FAM_QCONV 4          /* the k = 3 query convolution of both hands (16 % of the MACs)                                    */
#define EV2H_FAM_DENSE 8          /* the other dense layers: sa3, fp3, fp2, the regressors' sa2                                      */
#define EV2H_FAM_ALL 15
#define EV2H_W_EQUALIZED 1        /* the hidden channels were equalised by ev2h_pack_weights (the F16X2 accuracy contract above)  */
#define EV2H_W_UNEQUALIZED_OK 2   /* the caller knows its F16X2 weights are not equalised and wants them run anyway               */

/* ---- weight packing (host side): checkpoint -> the ev2h_weights a forward takes ---------------------------------------------
 * Replaces what nn.Module.load_state_dict + eval-mode BatchNorm do for the reference (model/model.py:14-23, demo.py:83-84,
 * pointnet2_utils.py:198,256,314, TEHNet.py:49-55,135-166): `tensors` are the checkpoint's entries as HOST arrays under the
 * reference's own names (a leading "module." is ignored, model.py:16-21; `num_batches_tracked` entries are ignored); the
 * schema is checked strictly (missing / unexpected key, wrong shape -> EV2H_ERR_ARG with the key in ev2h_last_error(), as
 * load_state_dict(strict=True) raises).  The packer
 *   1. folds eval-mode BatchNorm in float64 (Conv -> BN -> ReLU into W, b; Conv/Linear -> ReLU -> BN kept as a post-ReLU affine
 *      except in the segmentation head, where it folds forward into the last k=1 convolution; Conv -> BN of the second query
 *      convolution into W, b), rounded to fp32 once;
 *   2. with EV2H_PACK_EQUALIZE multiplies every hidden channel by e_c = 2^round(log2 sqrt(|consumer columns c| / |producer
 *      row c|)) at its producers and divides its consumers' columns by e_c (three sweeps; hidden tensors that share a
 *      contraction with raw coordinates are brought level with those): exact powers of two, so the network function and every
 *      fp32 partial sum are unchanged up to an exact scaling, and the F16X2 planes see well-conditioned operands;
 *   3. lays the weights out as the kernels read them (layer-1 feature weights of all radius branches stacked, W2 rows padded
 *      to 32, W3 columns to 8, group-all inputs as [features | xyz | 0], k=3 taps tap-major), splits them into the operand
 *      planes of `precision` (ev2h_tile_geometry) and computes the F16X2 range bounds;
 *   4. uploads everything into ONE device allocation on the current device, owned by the returned handle.
 * The handle owns the memory ev2h_packed_weights() points into: free it only when no forward that uses it is in flight (and no
 * captured graph refers to it). */
typedef struct ev2h_tensor_desc {
    const char* name;             /* checkpoint key                                                     */
    const void* data;             /* HOST pointer, contiguous, row-major                                */
    int dtype;                    /* EV2H_DT_*                                                          */
    int ndim;
    int64_t shape[4];
} ev2h_tensor_desc;
#define EV2H_DT_F32 0
#define EV2H_DT_F64 1
#define EV2H_DT_I64 2             /* accepted for num_batches_tracked only */
#define EV2H_PACK_EQUALIZE 1      /* step 2 above; what every F16X2 user wants                                                   */
#define EV2H_PACK_HOST_ONLY 2     /* no device: the weights view points into the handle's HOST copy (layout tests without a GPU)  */
#define EV2H_PACK_UNEQUALIZED_OK 4 /* sets EV2H_W_UNEQUALIZED_OK in the view (explicit opt-out of the F16X2 contract)             */
/* [ABI 8] precision == EV2H_PREC_F16 only: which kernel families run on one fp16 plane (bits 8..11 of flags = an EV2H_FAM_* mask;
 * the others are packed and run as F16X2).  No bits set = EV2H_FAM_ALL, the mode's definition; a partial mask is the "mixed" form
 * (e.g. EV2H_PACK_F16_FAMILIES(EV2H_FAM_SA): only the fused set abstractions reduced). */
#define EV2H_PACK_F16_FAMILIES(mask) (((mask) & 15) << 8)
typedef struct ev2h_packed ev2h_packed;
int ev2h_pack_weights(const ev2h_tensor_desc* tensors, int n, int in_channels, int precision, int flags, ev2h_packed** out);
void ev2h_packed_free(ev2h_packed* p);
const ev2h_weights* ev2h_packed_weights(const ev2h_packed* p);
size_t ev2h_packed_bytes(const ev2h_packed* p);                 /* size of the device allocation */
/* Introspection for tests and tools: the i-th packed array (0 <= i < ev2h_packed_tensor_count): name, [rows][cols], element type
 * (EV2H_DT_F32, or -1 = bytes of a plane image), its HOST copy and its device address (NULL with EV2H_PACK_HOST_ONLY). */
int ev2h_packed_tensor_count(const ev2h_packed* p);
int ev2h_packed_tensor(const ev2h_packed* p, int i, const char** name, int* rows, int* cols, int* dtype, const void** host,
                       const void** device);
/* Equalisation factors of the i-th hidden tensor (0 <= i < ev2h_packed_equalization_count; 0 tensors without EV2H_PACK_EQUALIZE):
 * e [n] powers of two, the accumulated factor channel c was multiplied by. */
int ev2h_packed_equalization_count(const ev2h_packed* p);
int ev2h_packed_equalization(const ev2h_packed* p, int i, const char** name, const double** e, int* n);
/* F16X2 only (0 entries otherwise): how the i-th weight matrix sits inside its one power-of-two scale -- counts = {non-zero weights,
 * weights with 0 < |w / u| < 2^-3 (more than ~2^17 below the matrix maximum: their low fp16 plane is subnormal), weights below 2^-14}.
 * The weight side of the accuracy contract at the top of this file.  Measured on synthetic checkpoints (4.6 M weights): with
 * EV2H_PACK_EQUALIZE ~150 weights below 2^-17 for a plain checkpoint and for one whose channels were rescaled to 2^32 apart; 1 % for
 * log-normal heavy-tailed weights (tiny weights next to outliers 400 x the median: they do not matter); WITHOUT equalisation 66 % of
 * the weights of the rescaled checkpoint (the 0.23 relative error quoted above). */
int ev2h_packed_weight_spread_count(const ev2h_packed* p);
int ev2h_packed_weight_spread(const ev2h_packed* p, int i, const char** name, uint64_t counts[3]);
/* The image builders on their own (operator-level callers of ev2h_sa_mlp_max / ev2h_fp_mlp / ev2h_gemm): W2 [C2][C1], W3 [C3][C2],
 * W [N][K] row-major float64 on the HOST; img* = caller's HOST buffers of ev2h_pack_*_bytes bytes; u* = the power-of-two factor
 * the planes were divided by (w2_unscale / w3_unscale / w_unscale).  planes: 1 BF16, 2 F16X2, 3 BF16X3, 4 F16 (one fp16 plane of W / u). */
int ev2h_pack_sa_image_bytes(int C1, int C2, int C3, int planes, size_t out[2]);
int ev2h_pack_sa_images(const double* W2, const double* W3, int C1, int C2, int C3, int planes, void* img2, void* img3, float* u2,
                        float* u3);
size_t ev2h_pack_gemm_image_bytes(int N, int K, int planes, int tile_rows);
int ev2h_pack_gemm_image(const double* W, int N, int K, int planes, int tile_rows, void* img, float* u);
float ev2h_plane_unscale(const double* W, size_t count, int planes);

typedef struct ev2h_outputs {
    float* class_logits;          /* [B,4,N]                                                           */
    float* params[2];             /* [B][P] left, right; P = 3 + n_pose_params + 10 + 3: the checkpoint's head width (22)  */
    float* vertices[2];           /* [B][778][3]                                                       */
    float* joints[2];             /* [B][21][3]                                                        */
    /* Floats between consecutive WINDOWS of each output (0 = dense: 4N, 22, 2334, 63).  With all four set to one row width the
     * forward writes a window's predictions side by side into one row of a caller-owned [B][row] matrix -- e.g. straight into
     * this rank's slice of the all-gather buffer (ev2hands_amd/dist.py: [4N logits | left 22 + 2334 + 63 | right ...]), so that
     * the multi-GPU path needs no packing copy. */
    size_t logits_stride, params_stride, vertices_stride, joints_stride;
} ev2h_outputs;

size_t ev2h_workspace_bytes(int B, int N);
/* TEHNet.forward, model/TEHNet.py:168-197, for B windows of N points with C channels.
 * fps_init: device int64 [4][B] in the reference's RNG consumption order (enc.sa1, enc.sa2,
 * left.sa1, right.sa1).  workspace: device buffer of at least ev2h_workspace_bytes(B, N).
 * Size contract: 128 <= N <= 32768 points per window, B limited by the workspace only.  Below 128 the reference itself fails
 * (query_ball_point indexes an N-column tensor with an nsample = 128 column mask, pointnet2_utils.py:103-106: IndexError).  It has
 * no upper limit, but materialises [B, S, N] int64 index tensors (8.4 MB per window and stage at N = 2048, 134 MB at 32768).  Up to
 * 8192 points (BASELINE.json's dense-window configuration, 4x the reference's operating point) the selection kernels keep a window's
 * points in registers / LDS; from 8193 to 32768 they run 1024-thread / global-memory variants (same arithmetic, same results).
 * ev2h_event_window_build takes at most 32768 raw events per window (LDS sort; the reference's windows are 2048 events,
 * erpc.py:170, or a few thousand, evaluation_stream.py:124-146).
 * Non-finite inputs or weights are outside the contract ("garbage in, garbage out", as in the reference), but the garbage differs: the
 * integer-max ReLU of the 16-bit modes maps a NaN with the sign bit set to 0 where torch.relu propagates it.
 * mano_left / mano_right may be NULL: that hand's MANO layer is skipped (out->vertices / joints of the hand are not touched)
 * and the caller applies its own hand model to out->params, as TEHNet.py:103 allows any callable. */
int ev2h_forward(const ev2h_weights* w, const ev2h_mano_consts* mano_left, const ev2h_mano_consts* mano_right,
                 float* xyz_cm, int B, int C, int N, int mhlnes, const int64_t* fps_init, const ev2h_outputs* out,
                 void* workspace, size_t workspace_bytes, ev2h_stream_t stream);

/* F16X2 spread report (the activation side of the accuracy contract at the top of this file).  After an ev2h_forward with F16X2
 * weights, for every operand tensor of a contraction that is materialised in `workspace` (names: ev2h_range_report_entries; the
 * hidden layers inside the fused kernels never reach memory and are not listed) and every window b:
 *   counts[(i * B + b) * 3 + 0] = non-zero values,
 *   counts[(i * B + b) * 3 + 1] = values with 0 < |v| * s < 2^-3  -- more than ~2^17 below the window's maximum: their low fp16 plane is
 *                                 subnormal, fewer than 22 bits survive the split (absolute error <= 2^-39 of the maximum),
 *   counts[(i * B + b) * 3 + 2] = values with 0 < |v| * s < 2^-14 -- more than ~2^28 below the maximum: the high plane is subnormal too,
 * with s the power of two the consumer scales the window by (from the tensor's range record(s)).  counts: DEVICE uint32
 * [entries][B][3], zeroed here.  A separate pass over the workspace of the LAST forward (same B, N): nothing is added to the
 * forward itself.  Large fractions in column 1 say that this input / checkpoint leans on values the F16X2 split resolves less
 * finely than fp32 does -- compare with BF16X3 (TEHNet.verify_precision does) before trusting the mode. */
int ev2h_range_report_entries(const char** names, int max_names);     /* returns the number of entries */
int ev2h_range_report(void* workspace, int B, int N, uint32_t* counts, ev2h_stream_t stream);

/* Measurement hook (bench.py): record caller-owned hipEvent_t pairs around ONE launch site of
 * ev2h_forward, on the forward's stream.  tag = "<module>.<branch>" with module in {sa1, sa2, manoL,
 * manoR} (the fused set-abstraction kernels), "fp1" (the fused feature-propagation chain) or "qconv0" (the k = 3 query
 * convolution GEMM of both hands, TEHNet.py:150-153).  Call i uses pair i % n.  tag == NULL disables. */
int ev2h_profile_set(const char* tag, void** start_events, void** stop_events, int n);

/* Debug access for parity tests: after ev2h_forward, device pointer of a named internal buffer in
 * `workspace` (e.g. "fps1", "gidx1_0", "l1cat", "l0", "sim", "hf8") and its element count; NULL if unknown.
 * "rng.<tensor>" (e.g. "rng.l0", "rng.p1b") = the F16X2 range record of that tensor, "p1scale" = the storage scales of the
 * five layer-1 tables (enc.sa1, enc.sa2, left, right, fp1; only enc.sa2 and fp1 are computed in the default F16X2 path) and [ABI 8], row
 * 5, of l0 when F16 stores it as fp16: float [6][B]. */
const void* ev2h_workspace_buffer(void* workspace, int B, int N, const char* name, size_t* count);
/* [ABI 8] The same with the element type: *elem_type = 0 for 4-byte elements (float32 / int32 / uint32, as documented per buffer),
 * 1 for bf16, 2 for fp16 TIMES the window's power of two ("p1scale" row 5: float [6][B]) -- `count` then still counts VALUES, stored
 * 2 bytes each.  Today that is "l0" after a forward of the calling thread in BF16 (1) / F16 (2) mode with the fused fp1 /
 * segmentation-head / query-convolution forms (EV2H_L0_F32=1 keeps it float32): the one N-row, 256-wide tensor of the path is then
 * written and read as 16-bit values; in F16 its range record "rng.l0" is the maximum of the STORED values.  A debugger that reads "l0" as float32 in that mode compares garbage
 * (ADVICE r5); ev2hands_amd's TEHNet.debug_buffer widens it. */
const void* ev2h_workspace_buffer_ex(void* workspace, int B, int N, const char* name, size_t* count, int* elem_type);

#ifdef __cplusplus
}
#endif
#endif /* EV2HANDS_HIP_H */
