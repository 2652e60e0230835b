// Internal launch interfaces shared by the .hip translation units of libhvn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel instantiation, device), from any
// host thread.  `done` = one bit per device ordinal (function-local static of the launcher).  Returns 0, or -2 when the runtime refuses.
static inline int hvn_max_lds_once(const void *kern, int bytes, std::atomic<unsigned long long> &done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -2;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

struct ConvArgs {
    const float *x;
    long xsn, xsy, xsx;
    int H, W, Cin;
    const float *w;
    const float *bias, *pre_s, *pre_b, *post_s, *post_b;
    const float *res;
    long rsn, rsy, rsx;
    float *y;
    long ysn, ysy, ysx;
    int N, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, relu;
    long M;
    long m_tiles;
    int n_tiles;
    int groups;
    const float *x2;      // optional second 1x1 input (channels appended to the reduction)
    long x2sn, x2sy, x2sx;
    int Cin2, stride2;
    int nbatch;           // > 1: blockIdx.y selects one of nbatch independent problems
    int stagger;          // experiment switch, see hvn_conv.hip
    unsigned long long *dbg;  // optional: per-workgroup {start, k-loop end, epilogue end, HW_ID | LDS base << 32} (tools/conv_trace.py)
    long xb, wb, yb;      // element strides between them
    // EXPERIMENT (HVN_EXP_BLOCKED, timing only -- profiles/r02_experiments.md section 9): channel-BLOCKED addressing [C/32][H][W][32]:
    // xsb = element stride between 32-channel input blocks (launch_conv sets BK = channels-last when 0); ysb / rsb likewise for the
    // output / residual (0 = channels-last)
    long xsb, ysb, rsb;
    int blk_shift;        // log2 of the channel block of the experiment (5 or 7); 0 = off
};

int hvn_launch_conv(const ConvArgs &a, int tile_n, hipStream_t stream);

// Two chained 1x1 convs of a residual block in one launch (hvn_conv_chain.hip):
//   y  = [relu(. * post_s + post_b)]( W1 . x (+ W1b . x2) + res )        C channels        (a unit's conv3 + shortcut)
//   y2 = [relu]( W2 . a + bias2 ),  a = relu(y * pre_s + pre_b) or y     N2 channels       (the next unit's conv1)
#ifndef HVN_CONV_XCD_CONTIG
#define HVN_CONV_XCD_CONTIG 1      // (lib.py VARIANTS "noxcd": 0, the A/B build)
#endif
// pixel tile of workgroup (xcd = blockIdx % 8, seq_m) of a convolution launch with `groups` = ceil(m_tiles / 8) tiles per XCD
static __device__ __forceinline__ int hvn_m_tile(int xcd, int seq_m, int groups, bool contiguous)
{
    return (HVN_CONV_XCD_CONTIG && contiguous) ? xcd * groups + seq_m : seq_m * 8 + xcd;
}

struct ChainArgs {
    const float *x;       // conv3 input view [N][Ho][Wo][K1]
    long xsn, xsy, xsx;
    int K1;
    const float *x2;      // optional second input (the strided 1x1 shortcut's), sampled at (oy * stride2, ox * stride2)
    long x2sn, x2sy, x2sx;
    int K1b, stride2;
    const float *w1;      // [C_pad][(K1 + K1b) / 32][1][32]
    const float *res;     // optional residual view (may alias y)
    long rsn, rsy, rsx;
    float *y;
    long ysn, ysy, ysx;
    int C;
    const float *post_s, *post_b, *pre_s, *pre_b;   // optional per-channel affines [C]
    const float *w2;      // [N2_pad][C / 32][1][32]
    const float *bias2;   // optional [N2]
    int relu2;
    float *y2;
    long y2sn, y2sy, y2sx;
    int N2;
    int N, Ho, Wo;
    long M;
    int bm;               // pixels per workgroup: 64, or anything else for 128
    unsigned long long *dbg;   // optional (HVN_CHAIN_TRACE): 10 cycle stamps per workgroup, see tools/chain_bench.py
};
int hvn_launch_conv_chain(const ChainArgs &a, hipStream_t stream);
int hvn_chain_supported(int c, int n2);
// the same chain for the bf16 path (hvn_conv_chain_bf16.hip): bf16 views (strides in elements), hvn_conv_bf16.hip's weight packing, 64 pixels per workgroup
int hvn_chain_bf16_supported(int k1, int k1b, int c, int n2);
int hvn_launch_conv_chain_bf16(const ChainArgs &a, hipStream_t stream);
// the same op with w1 / w2 = bf16 planes of the fp32 packings and both GEMMs' products on the bf16 matrix pipe (hvn_conv_chain_x3.hip)
int hvn_launch_conv_chain_x3(const ChainArgs &a, int terms, hipStream_t stream);
// the same op, same bits, conv3's input tile resident in registers and every other operand a chunk ahead in flight (hvn_conv_chain_x3r.hip);
// exists for K1 = 64 (+ K1b = 64 without a residual view and with N2 = 64)
int hvn_chain_x3r_supported(const ChainArgs &a);
int hvn_launch_conv_chain_x3r(const ChainArgs &a, int terms, hipStream_t stream);
int hvn_launch_conv_bf16(const ConvArgs &a, int tile_n, hipStream_t stream);   // x, res, y, x2, w are bf16; bias / scales fp32
// the same convolution (same packing, same bits) with both operands staged by LDS-DMA: bm = 256 | 128 pixels x 128 channels (hvn_conv_bf16g.hip)
int hvn_launch_conv_bf16g(const ConvArgs &a, int bm, hipStream_t stream);
// fp32 in / out; w = [cout_pad][k-step][3][32] bf16 planes of the fp32 weights; terms = 9 | 6 partial products (hvn_conv_x3.hip)
int hvn_launch_conv_x3(const ConvArgs &a, int tile_n, int terms, hipStream_t stream);
// the same convolution (same packing, same bits) with both operands staged by LDS-DMA, bm = 256 | 128 pixels x 128 channels (hvn_conv_x3g.hip)
int hvn_launch_conv_x3g(const ConvArgs &a, int bm, int terms, hipStream_t stream);
int hvn_conv_x3g_supported(const ConvArgs &a, int bm);
// fp32 packings ([...][32]-float granules) -> their bf16 planes [3][32] per granule (training: weights change every step)
int hvn_launch_split_x3(const float *src, uint16_t *dst, long granules, hipStream_t stream);

struct Conv0Args {
    const void *img;      // uint8 or float32
    long isn, isy, isx, isc;  // strides in elements
    int is_f32;
    int H, W;             // input extent
    const float *w;       // [7][7][3][64]
    const float *bias;    // [64]
    float *y;
    long ysn, ysy, ysx;
    int N, Ho, Wo, pad;
    int relu;             // 1: ReLU after the bias (inference, BN folded); 0: raw conv output (training)
    int out_bf16;         // y is bf16 (cfg 3) instead of fp32
};
int hvn_launch_conv0(const Conv0Args &a, hipStream_t stream);

struct UpAddArgs {
    const float *lo;
    long lsn, lsy, lsx;
    const float *skip;
    long ssn, ssy, ssx;
    float *y;
    long ysn, ysy, ysx;
    int N, H, W, C;  // output extent
    int bf16;        // all three tensors are bf16
};
int hvn_launch_upadd(const UpAddArgs &a, hipStream_t stream);

struct HeadArgs {
    const float *x;
    long xsn, xsy, xsx;
    const float *w;     // [cout][64]
    const float *bias;  // [cout]
    float *y;           // NCHW [N][cout][H][W]
    int N, H, W, Cout;
    int in_bf16;        // x is bf16 (the logits stay fp32)
};
int hvn_launch_head(const HeadArgs &a, hipStream_t stream);

int hvn_launch_extract_patches(const uint8_t *img, int H, int W, const int32_t *coords, int P, int win, int pad_t, int pad_l, uint8_t *out,
                               hipStream_t stream);

struct PredMapArgs {
    const float *tp, *np, *hv;  // NCHW logits
    float *y;                   // [N][H][W][3|4]
    int N, H, W, nr_types;      // nr_types = 0: no type channel
};
int hvn_launch_predmap(const PredMapArgs &a, hipStream_t stream);

struct WinoArgs {
    const float *x;       // WINO_IN: input view; WINO_OUT: M [n*n][tiles][C] per sample (n = m + 4)
    long xsn, xsy, xsx;
    float *y;             // WINO_IN: V [n*n][tiles][C] per sample; WINO_OUT: output view
    long ysn, ysy, ysx;
    const float *mat;     // B^T (n x n) or A^T (m x n)
    const float *bias;    // WINO_OUT only
    int N, H, W, C;       // WINO_IN: input window extent / channels; WINO_OUT: output extent / cout
    int ty, tx, pad, relu;
    int m;                // output tile edge: 2 or 4
    int r;                // filter edge: 5 or 3 (n = m + r - 1)
    int accum;            // WINO_OUT: y += result (data gradients accumulate)
    const float *lo;      // WINO_IN, optional: the input is nearest2x(lo) + x (net_utils.py:284-294 UpSample2x + the skip add), formed on
    long lsn, lsy, lsx;   // the fly: lo is the half-resolution view, x the full-resolution skip; nullptr = plain input
};
int hvn_launch_wino_in(const WinoArgs &a, hipStream_t stream);
int hvn_launch_wino_out(const WinoArgs &a, hipStream_t stream);

// ---- training step (hvn_train.hip) -------------------------------------------------------------
struct PackArgs {
    const float *src;     // parameter layout [cout][kh*kw][cin_g]
    float *dst;
    int cout, cin_g, groups, taps;
    int mode;             // 0 forward [lead_pad][cin/32][taps][32]; 1 dgrad [lead_pad][cout/32][taps][32] (transposed, taps flipped); 2 conv0
    int lead_pad;         // padded leading dimension (multiple of the conv kernel's column tile)
    const float *gmat;    // modes 3 / 4 (Winograd F(4x4,5x5) transforms U = G g G^T, forward / data-gradient): G [8][5]
};
int hvn_launch_pack_w(const PackArgs &a, hipStream_t stream);
// every mode 0 / 1 / 2 packing of a step in one launch: device table + first workgroup of each packing (first_block[n] = grid size)
int hvn_launch_pack_w_multi(const PackArgs *tbl, const int *first_block, int n, long blocks, hipStream_t stream);

struct WgradArgs {
    const float *x;       // conv input view
    long xsn, xsy, xsx;
    int H, W, Cin;
    const float *dy;      // output-gradient view
    long dsn, dsy, dsx;
    int Ho, Wo, Cout;
    float *dw;            // [Cout][KH*KW][Cin_g], accumulated with atomics
    int N, KH, KW, stride, pad_t, pad_l, groups, Cin_g;
    int tiles_m, tiles_n;
    unsigned rows_per_split;
    int nbatch;           // > 1: blockIdx.z selects one of nbatch independent problems
    long xb, db, wb;      // element strides of x, dy, dw between them
    int want_wgs;         // workgroups the K split aims at (0: the default)
    // deterministic reduce (hvn_run_train_plan_ws): split s of the pixel sum STORES its tile into copy s of the gradient tensor in
    // `part` (part_stride floats per copy, part_cap floats in all) and hvn_launch_reduce_parts adds the copies to dw in split order
    float *part;
    long part_stride, part_cap;
};
int hvn_launch_wgrad(const WgradArgs &a, hipStream_t stream);
// the K split both weight-gradient launchers make: -> number of splits, *rows_per_split (a multiple of 32)
long hvn_wgrad_split(const WgradArgs &a, long tiles, unsigned *rows_per_split);
// floats of `part` a launch of this shape needs (0: a single split, which needs none)
long hvn_wgrad_part_floats(const WgradArgs &a, int x3);
// dst[e] += part[0][e] + part[1][e] + ... in a fixed order (4 interleaved chains of ascending s, then chain 0 + 1 + 2 + 3)
int hvn_launch_reduce_parts(float *dst, const float *part, long elems, long nparts, long stride, hipStream_t stream);
// the same sum with its products on the bf16 matrix pipe from bf16x3 splits of both operands (hvn_wgrad_x3.hip): 128 x 128 channel tiles
int hvn_wgrad_x3_supported(const WgradArgs &a);
int hvn_launch_wgrad_x3(WgradArgs a, int terms, hipStream_t stream);
int hvn_launch_wino_dy(const struct WinoArgs &a, hipStream_t stream);
int hvn_launch_wino_dw(const float *du, float *dg, const float *gmat, int cout, int cin, hipStream_t stream);

#define HVN_BN_MAX_PARTS 256
struct BnArgs {
    const float *z;       // conv output (pre-normalisation)
    long zsn, zsy, zsx;
    const float *a;       // backward: the forward output relu(bn(z)) -- NOT READ since round 6 (the ReLU mask is recomputed from z, save[0..2C))
    float *a_out;         // forward: where it is written
    long asn, asy, asx;
    const float *da;      // backward: gradient of a
    long gsn, gsy, gsx;
    float *dz;            // backward: gradient of z (accumulated), may be null
    long dsn, dsy, dsx;
    double *ws;           // [HVN_BN_MAX_PARTS][2*C] partial sums (no initialisation needed)
    float *save;          // [4*C] scale, shift, mean, rstd
    float *coef;          // [3*C] backward coefficients
    const float *gamma, *beta;
    float *dgamma, *dbeta, *running_mean, *running_var;
    int N, H, W, C, lq, nparts;
    float eps, momentum;
    int dz_store;         // backward: dz = ... instead of dz += ... (the caller knows this launch is the first writer of dz in the step)
};
int hvn_launch_bn_forward(BnArgs a, hipStream_t stream);
int hvn_launch_bn_backward(BnArgs a, hipStream_t stream);

struct UpAddBwdArgs {
    const float *dy;
    long ysn, ysy, ysx;
    float *dlo;           // may be null
    long lsn, lsy, lsx;
    float *dskip;         // may be null
    long ssn, ssy, ssx;
    int N, H, W, C;       // extent of dy
};
int hvn_launch_upadd_bwd(const UpAddBwdArgs &a, hipStream_t stream);

struct HeadBwdArgs {
    const float *x;       // head input a [N][H][W][64] view
    long xsn, xsy, xsx;
    float *dx;            // its gradient (accumulated)
    long dsn, dsy, dsx;
    const float *dl;      // logit gradient NCHW [N][Cout][H][W]
    const float *w;       // [Cout][64]
    float *dw, *db;
    int N, H, W, Cout;
    float *part;          // deterministic reduce: [workgroups][Cout * 64 + Cout] partial sums (NULL: fp32 atomics)
    long part_cap;
};
int hvn_launch_head_bwd(const HeadBwdArgs &a, hipStream_t stream);
long hvn_head_bwd_part_floats(const HeadBwdArgs &a);

struct Conv0WgradArgs {
    const uint8_t *img;
    long isn, isy, isx;
    int H, W;
    const float *dy;
    long ysn, ysy, ysx;
    float *dw;            // [64][7][7][3]
    int N, Ho, Wo, pad;
    float *part;          // deterministic reduce: [workgroups][64 * 147] partial sums (NULL: fp32 atomics)
    long part_cap;
};
int hvn_launch_conv0_wgrad(const Conv0WgradArgs &a, hipStream_t stream);
long hvn_conv0_wgrad_part_floats(const Conv0WgradArgs &a);

struct LossArgs {
    const float *l_np, *l_hv, *l_tp;   // NCHW logits
    const int32_t *t_np, *t_tp;        // [N][H][W]
    const float *t_hv;                 // [N][H][W][2]
    float *d_np, *d_hv, *d_tp;         // logit gradients (stage 1)
    double *sums;                      // [64]
    float *gws;                        // [N][H][W][2] focus-weighted Sobel differences
    int N, H, W, T;
    double m_total;
    float wt[6];                       // np bce, np dice, hv mse, hv msge, tp bce, tp dice
    double *parts;                     // deterministic reduce: [workgroups][64] per-workgroup sums (NULL: double atomics into sums)
    long parts_cap;                    // doubles
};
long hvn_loss_part_doubles(int n, int h, int w);
int hvn_launch_loss(const LossArgs &a, int stage, hipStream_t stream);
int hvn_launch_adam(float *w, const float *g, float *m, float *v, long n, float b1, float b2, float eps, float step_size,
                    float inv_bc2_sqrt, hipStream_t stream);

struct hvn_op;
int hvn_internal_run_one(const hvn_op *op, int batch, hipStream_t s);

// ---- training targets (hvn_targets.hip) ---------------------------------------------------------
struct hvn_aug_sample;   // include/hvn.h
int hvn_launch_aug_shape(const uint8_t *img, const int32_t *ann, int h, int w, int c, const struct hvn_aug_sample *prm, int n, int oh, int ow,
                         uint8_t *oimg, int32_t *oann, hipStream_t stream);
int hvn_launch_aug_input(const uint8_t *src, const struct hvn_aug_sample *prm, const float *noise, int n, int h, int w, uint8_t *dst, hipStream_t stream);
size_t hvn_targets_ws_bytes(int n, int h, int w);
int hvn_launch_gen_targets(const int32_t *ann, int n, int h, int w, int ch, int cw, float *hv, int32_t *np_map, void *ws, size_t ws_bytes,
                           hipStream_t stream);
