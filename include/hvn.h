/*
 * hvn.h -- C ABI of libhvn_hip.so, the MI355X (gfx950) HoVer-Net hot path.
 *
 * The reference (vqdang/hover_net) is pure Python and has no FFI; its boundary for
 * this path is three duck-typed callables looked up by module name
 * (/root/reference/infer/base.py:56-78).  Each entry point below names the
 * reference interface it stands behind.  All pointers marked `dev` are device (HBM)
 * addresses owned by the caller; `stream` is a hipStream_t passed as void*; every
 * function returns 0 on success or a negative hvn_status and never throws.  One host
 * thread per stream.  See INTEGRATION.md for the ctypes binding.
 */
#ifndef HVN_H
#define HVN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVN_API __attribute__((visibility("default")))

typedef enum hvn_status {
    HVN_OK = 0,
    HVN_E_ARG = -1,      /* malformed descriptor (alignment, channel multiple, kind) */
    HVN_E_LAUNCH = -2,   /* hipLaunch / hipGetLastError failure (hvn_last_error() has the text) */
    HVN_E_NOGPU = -3,    /* no gfx950 device visible */
    HVN_E_SIZE = -4      /* workspace too small */
} hvn_status;

/* Strided view: element (n,y,x,c) lives at base[n*sn + y*sy + x*sx + c*sc].  Every kernel but
 * CONV0 requires channels-last (sc = 1; 0 is read as 1). */
typedef struct hvn_view {
    void   *base;        /* dev */
    int64_t sn, sy, sx;  /* strides in elements */
    int32_t h, w, c;     /* window extent */
    int32_t sc;          /* channel stride in elements (CONV0 input only, e.g. h*w for NCHW) */
} hvn_view;

enum { HVN_OP_CONV0 = 1, HVN_OP_CONV = 2, HVN_OP_UPADD = 3, HVN_OP_HEAD = 4, HVN_OP_PREDMAP = 5, HVN_OP_WINO_IN = 6, HVN_OP_WINO_OUT = 7, HVN_OP_CHAIN = 8 };

/*
 * One fused launch of the network plan (hover_net_amd/plan.py lowers
 * /root/reference/models/hovernet/net_desc.py:101-145 to an array of these).
 *   CONV0   x = image, uint8 (x_dtype 0; infer_step hands NHWC bytes) or float32 0..255 (x_dtype 1;
 *           HoVerNet.forward's NCHW contract), w = [kh][kw][3][64] taps (1/255 and BN folded), bias, relu
 *   CONV    y = epi( conv( pro(x) ) ):  pro = relu(x*pre_scale+pre_shift) if pre_scale,
 *           w = [cout_pad][x.c/32][kh*kw][32] fp32 (cout_pad = multiple of tile_n; reduction order =
 *           32-channel slab, tap, channel); groups = 4 declares the block-diagonal packing of a grouped
 *           conv with 32 input / 8 output channels per group (dense-unit conv2, net_utils.py:114-125):
 *           the kernel then multiplies each slab only against its own group's output columns;
 *           x2 (1x1 convs): a second input tensor whose channels are appended to the reduction, read at
 *           (oy*s2, ox*s2) with s2 = `_rsv` -- y = W1.x + W2.x2 fuses a residual block's strided 1x1 shortcut
 *           (net_utils.py:229-230) into its first conv3, the weight rows being [x.c | x2.c] wide,
 *           epi = (+bias) (relu) (+res) (relu(.*post_scale+post_shift) if post_scale)
 *           nbatch > 1: `nbatch` independent problems in one launch; problem b reads x.base + b*batch_stride[0],
 *           w + b*batch_stride[1] and writes y.base + b*batch_stride[2] (elements) -- the n*n transform-domain
 *           products of a Winograd convolution,
 *   WINO_IN  Winograd F(m x m, r x r) input transform of a stride-1 conv: F(2,5) / F(4,5) for the 5x5 decoder convs
 *           (net_desc.py:45,52,59 conva), F(4,3) for the encoder's 3x3 convs (net_utils.py:186-196), n = m + r - 1:
 *           x = input view (zero padding pad_t/pad_l), y = V as [n*n][tiles][c] per sample (y.w = tiles),
 *           kh x kw = tile grid, w = B^T (n x n), stride = m, _rsv = r (both 0: r = 5 and m from y.h = 36 | 64),
 *           res (optional): a half-resolution view; the transformed input is then nearest2x(res) + x, i.e. the UpSample2x + skip add
 *           of net_utils.py:284-294 / net_desc.py:133-143 formed on the fly (x = the skip), bit-identical to UPADD followed by WINO_IN,
 *   WINO_OUT y = A^T M A (+bias)(relu): x = M as [n*n][tiles][cout] per sample, w = A^T (m x n), kh x kw = tile grid
 *           covering y (a partial last tile's surplus outputs are dropped); res = y: accumulate (y += ...),
 *   CHAIN   two chained 1x1 convs of a residual block (net_utils.py:250-266) in one launch, the second running on the first's
 *           output while it is still on chip:   y  = post( W.x (+ W'.x2) + res )      -- a unit's conv3 (+ fused shortcut), `cout` channels,
 *                                               y2 = relu( W2.a + bias2 ), a = relu(y*pre_scale+pre_shift) if pre_scale else y
 *           -- the next unit's pre-activation + conv1 (+ folded BN), `cout2` in {64, 128} channels; w / w2 packed like CONV
 *           ([cout_pad][(x.c + x2.c)/32][1][32] and [cout2_pad][cout/32][1][32]); NOTE pre_scale / pre_shift act on y here, not on x;
 *           res (optional, may alias y) must have y's strides; x.c + x2.c >= 64, cout % 64 == 0.  Bit-identical to the two CONV
 *           launches it replaces.  act_dtype 2 | 3: both GEMMs form their products on the bf16 matrix pipe from bf16x3 splits
 *           (csrc/hvn_conv_chain_x3.hip): w / w2 then hold the bf16 planes of the fp32 packings ([rows][k-step][3][32] bf16, see
 *           act_dtype below), 128 pixels per workgroup; bit-identical to the two CONV launches with the same act_dtype.  tile_n = 128 + 0x400
 *           (1152) of such a CHAIN selects the form with conv3's input tile resident in registers and every other operand a chunk
 *           ahead in flight (csrc/hvn_conv_chain_x3r.hip: x.c = 64, and x2.c = 64 with cout2 = 64 and no res; same bits; HVN_E_ARG otherwise).
 *   UPADD   y = nearest2x(x) + res
 *   HEAD    y.base = NCHW logits [n][cout][h][w];  w = [cout][64], bias[cout]
 *   PREDMAP y.base = [n][h][w][3|4] = [argmax(tp)?, softmax(np)[1], hv0, hv1]
 *           (run_desc.py:185-194); x.base = np logits, res.base = hv logits, w = tp logits or NULL,
 *           cout = nr_types (0 if none)
 */
typedef struct hvn_op {
    int32_t kind, kh, kw, stride, pad_t, pad_l, relu, cout, tile_n, x_dtype, groups, _rsv;
    hvn_view x, res, y;
    hvn_view x2;         /* CONV, 1x1 only: optional second input (base NULL = none), sampled with spatial stride `_rsv` */
    const float *w, *bias, *pre_scale, *pre_shift, *post_scale, *post_shift; /* dev */
    int64_t batch_stride[3]; /* CONV with nbatch > 1: element strides of x, w, y between problems */
    int32_t nbatch;
    int32_t act_dtype;   /* 0: fp32 activations / weights; 1: bf16 activations (x, res, y, x2) and packed weights
                            ([cout_pad][ceil(x.c/64)][kh*kw][64] bf16, zero-filled past x.c), fp32 accumulation, fp32 bias /
                            scales; CONV0 (bf16 output), CONV, UPADD, HEAD (bf16 input, fp32 logits) honour it.
                            CONV only -- 2 | 3: fp32 activations, accumulation and outputs exactly as 0, but the products run on the
                            bf16 matrix pipe from exact three-way bf16 splits of the fp32 operands (csrc/hvn_conv_x3.hip): `w` then
                            holds the three bf16 planes of the fp32 packing, [cout_pad][k-step][3][32] bf16 (k-step = (x.c/32 slab,
                            tap), the fp32 packing's order) and batch_stride[1] counts bf16 elements; 2 = all nine partial
                            products (the fp32 dot product in another summation order), 3 = the six that carry more than 2^-24
                            of a product.  tile_n of such a CONV: 128 | 64 = 128 pixels x tile_n channels per workgroup
                            (hvn_conv_x3.hip); 128 + 0x300 (896) | 128 + 0x200 (640) = 256 | 128 pixels x 128 channels with both operands
                            staged by LDS-DMA (csrc/hvn_conv_x3g.hip; cout >= 128; same bits as the other forms); a CONV with act_dtype 1
                            takes the same two codes for the LDS-DMA form of the bf16 convolution (csrc/hvn_conv_bf16g.hip: no
                            prologue, cout >= 128, same bits as tile_n 128) */
    /* CHAIN only: the second conv's output view, packed weights, bias (or NULL) and channel count */
    hvn_view y2;
    const float *w2, *bias2; /* dev */
    int32_t cout2, _rsv2;
} hvn_op;

/* -- library ---------------------------------------------------------------------- */
HVN_API int         hvn_version(void);
HVN_API const char *hvn_build_id(void);   /* id of the sources the binary was compiled from; the Python binding refuses a stale library */
HVN_API const char *hvn_last_error(void);
HVN_API int         hvn_device_ok(void);  /* 1 if a gfx950 device is current */

/* -- network: HoVerNet.forward (net_desc.py:101-145) + infer_step epilogue (run_desc.py:171-197) */
HVN_API int hvn_run_plan(const hvn_op *ops, int n_ops, int batch, void *stream);
/* single launches, used by the per-kernel parity tests */
HVN_API int hvn_run_op(const hvn_op *op, int batch, void *stream);
/* per-launch timing of the CONV kernel: hvn_profile_enable(1) resets the tally and brackets every
 * CONV launch of the following hvn_run_plan / hvn_run_op calls with hipEvents on their stream;
 * hvn_profile_conv_ms() synchronises and returns the summed milliseconds (<0 if nothing was recorded) */
HVN_API int    hvn_profile_enable(int on);
HVN_API double hvn_profile_conv_ms(void);
HVN_API int    hvn_profile_conv_launches(void);
/* the same measurements one by one, in launch order: fills out[0..min(cap, launches)) and returns that count */
HVN_API int    hvn_profile_conv_ms_list(double *out, int cap);

/* -- patch extraction: infer/tile.py:46-94 _prepare_patching (numpy "reflect" padding) + dataloader/infer_loader.py:59-72
 * img: dev uint8 [h][w][3] (the UNPADDED source image); coords: dev int32 [n_patches][2] = (y, x) top-left corners in the
 * padded frame (patch_info[:, :2]); out: dev uint8 [n_patches][win][win][3]; pad_t / pad_l = (win - step) / 2. */
HVN_API int hvn_extract_patches(const uint8_t *img, int h, int w, const int32_t *coords, int n_patches, int win,
                                int pad_t, int pad_l, uint8_t *out, void *stream);

/* -- instance separation: post_proc.py:26-90 __proc_np_hv ---------------------------- */
/* bytes of device workspace needed for `n` maps of h x w */
HVN_API size_t hvn_postproc_workspace_bytes(int n, int h, int w);
/* pred: dev float32 [n][h][w][c] with [p, h, v] at channels c0..c0+2 (c0 = 1 when a type
 * channel leads, post_proc.py:109-114); inst: dev int32 [n][h][w].  Labels equal the
 * reference's (scipy raster-order numbering of the marker components). */
HVN_API int hvn_postproc(const float *pred, int n, int h, int w, int c, int c0,
                         int32_t *inst, void *workspace, size_t workspace_bytes, void *stream);
/* stage taps for tests (any may be NULL): blb int32, dist float64 (negated blur), marker int32 */
HVN_API int hvn_postproc_taps(const float *pred, int n, int h, int w, int c, int c0, int32_t *inst,
                              int32_t *blb, double *dist, int32_t *marker,
                              void *workspace, size_t workspace_bytes, void *stream);
/* which replay the marker-controlled watershed (post_proc.py:88) of the LAST hvn_postproc call on `workspace` took, summed over its
 * n maps (waits for `stream`): out[0] components that hold a marker, out[1] replayed on the small LDS window, out[2] on the bitmap
 * window, out[3] on the HBM window, out[4] handed to the one-lane heap by a mixed-label marker tie, out[5] by a full frontier,
 * out[6] component heap replays, out[7] WHOLE-TILE replays (a tie the component replay could not prove harmless), out[8] maps
 * flagged with such a tie, out[9] largest component bounding box. */
HVN_API int hvn_postproc_stats(const void *workspace, size_t workspace_bytes, int n, int h, int w, long long out[10], void *stream);

/* -- per-instance table: post_proc.py:119-181 process() (bbox, centroid, type vote) ----- */
typedef struct hvn_inst_rec {
    int32_t label;              /* value in the inst map */
    int32_t area;
    int32_t rmin, rmax, cmin, cmax; /* get_bounding_box semantics (misc/utils.py:18-28): max is exclusive */
    double  sum_x, sum_y;       /* cv2.moments m10, m01 over the bbox crop (offsets re-added): centroid = sum/area */
    int32_t type;               /* majority type, runner-up if 0 (post_proc.py:170-177); -1 if no type channel */
    int32_t type_count;         /* pixels of that type: type_prob = type_count / (area + 1e-6) */
} hvn_inst_rec;
/* records: dev [n][max_inst], slot j describes label j+1 (area 0 = label absent; labels above
 * max_inst are ignored); counts: dev int32 [n] = number of labels present; pred may be NULL when
 * nr_types == 0, else its channel 0 is the type map (post_proc.py:109-112). */
HVN_API size_t hvn_instance_table_workspace_bytes(int n, int max_inst, int nr_types);
HVN_API int hvn_instance_table(const int32_t *inst, const float *pred, int n, int h, int w, int c,
                               int nr_types, hvn_inst_rec *records, int32_t *counts, int max_inst,
                               void *workspace, size_t workspace_bytes, void *stream);

/* -- whole-slide merge: infer/wsi.py:569-599 post_proc_normal_tile_callback, :602-677 post_proc_fixing_tile_callback and :51-60
 * _remove_inst on a DEVICE-resident int32 instance map [H][map_w] (tiles strictly in the reference's order: the id offset is the
 * running maximum id and the fix-up windows overlap).  pred_inst: dev int32 [h][w] local ids of the tile at (y0, x0).
 * normal:  window = pred_inst (+ off where > 0).
 * fixing:  old ids wholly inside the window (np.unique(roi)[1:] minus np.unique(edge)[1:] -- "[1:]" drops the smallest value as the
 *          reference does) are zeroed and listed in `removed` (count in counters[0]; counters[1] / [2] = smallest value on the window
 *          edge / in the window, counters[3] = smallest value of pred_inst: np.unique(pred_inst)[1:] drops it too); touching[i] = 1 for new ids i <= n_local that overlap a kept old instance (they are dropped); the
 *          other new ids are written with + off.  id_flags: dev int32 [2][cap] epoch-stamped tables (zeroed once; cap > every id in
 *          the map), epoch > 0 and increasing from call to call; removed: dev [removed_cap]; counters: dev [5] (counters[4] = window pixels whose id is >= cap, i.e. beyond the id tables: must be 0, the caller checks); touching: dev [n_local + 1]. */
HVN_API int hvn_wsi_merge_normal(int32_t *inst_map, int64_t map_w, int y0, int x0, int h, int w, const int32_t *pred_inst, int32_t off,
                                 void *stream);
HVN_API int hvn_wsi_merge_fixing(int32_t *inst_map, int64_t map_w, int y0, int x0, int h, int w, const int32_t *pred_inst, int32_t n_local,
                                 int32_t off, int32_t epoch, int32_t *id_flags, int64_t cap, int32_t *removed, int32_t removed_cap,
                                 int32_t *counters, uint8_t *touching, void *stream);

/* -- contours: cv2.findContours(crop, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0] of process() (post_proc.py:132-143)
 * HOST function (O(perimeter) per instance over its bbox crop): inst = host int32 [h][w]; recs = host records
 * (slots with area 0 are skipped); pts = host int32 [max_pts][2] as (x, y) in map coordinates;
 * offs = host int64 [n_rec + 1] prefix offsets into pts.  Returns the total number of points or <0. */
HVN_API long hvn_trace_contours(const int32_t *inst, int h, int w, const hvn_inst_rec *recs, int n_rec,
                                int32_t *pts, long max_pts, int64_t *offs);


/* -- training step: run_desc.py:12-109 train_step (forward in train() mode, losses utils.py:54-172, backward, Adam) --
 * A training step is two hvn_top lists (forward, backward; hover_net_amd/train_plan.py lowers the network to them)
 * around the two loss stages.  Weights live in the parameter layout [cout][kh*kw][cin_g] (torch channels_last);
 * gradients are ACCUMULATED into their destinations (the caller zeroes the gradient arena once per step).
 *   NET          net = one forward-plan op (CONV0 / CONV / UPADD / HEAD); convolutions of the forward pass and the
 *                data gradients (stride-1 conv of the -- possibly dilated -- output gradient, res = y to accumulate)
 *   PACK_W       p[0] = weights, p[1] = packed copy for the conv kernel; mode 0 forward ([lead_pad][cin/32][taps][32],
 *                groups expanded block-diagonally), 1 data-gradient (transposed, taps flipped:
 *                [lead_pad][cout/32][taps][32]), 2 conv0 ([7][7][3][64] x 1/255); 3 / 4 Winograd F(4x4,5x5) transform
 *                U = G g G^T of a 5x5 conv for the forward / data-gradient pass ([64][lead_pad][k/32][32], p[2] = G
 *                [8][5]); cout, cin_g, groups, kh, kw
 *   BN_FWD       a = relu(batchnorm_train(z)): x = z, y = a; p[0] = double ws[256][2c] (scratch for the partial sums),
 *                p[1] = save[4c] (scale, shift, mean, rstd), p[2] = gamma, p[3] = beta, p[4] = running_mean,
 *                p[5] = running_var (updated: momentum, unbiased variance); eps, momentum
 *   BN_BWD       x = z, y = a (not read: the ReLU mask a > 0 is recomputed from z and save's scale / shift with the forward's own
 *                instruction), dy = grad a, dx = grad z (+=; mode & 1: = -- the caller knows this launch is the first writer of
 *                grad z in the step; base NULL: none); p[0] = ws, p[1] = save (as the BN_FWD of this step left it), p[2] = gamma,
 *                p[3] = grad gamma (+=), p[4] = grad beta (+=), p[5] = coef[3c] scratch
 *   WGRAD        p[0][cout][kh*kw][cin_g] += sum_pixels dy (x) x: x = conv input view, dy = output-gradient view;
 *                kh, kw, stride, pad_t, pad_l, groups; mode = workgroups the split of the pixel sum aims at (0: default;
 *                a tuning hint, the result is the same sum in another order); `_pad` = 6 | 9: the products run on the bf16 matrix
 *                pipe from exact three-way bf16 splits of both fp32 operands (csrc/hvn_wgrad_x3.hip; 9 = every partial product, 6 =
 *                those above 2^-24 of a product) where the shape has that form (ungrouped, cout >= 128, x.c % 128 == 0), 0 = fp32 pipe
 *   CONV0_WGRAD  x = uint8 image view, dy = grad of the conv0 output, p[0] = grad [64][7][7][3] (+=); pad_t
 *   UPADD_BWD    dy = grad of nearest2x(lo) + skip; dx = grad lo (+= 2x2 sums, nullable), y = grad skip (+=, nullable)
 *   HEAD_BWD     x = head input [h][w][64], dx = its grad (+=), p[0] = logit grad NCHW, p[1] = W [cout][64],
 *                p[2] = grad W (+=), p[3] = grad bias (+=); cout
 *   WGRAD with nbatch > 1: nbatch independent problems, problem b at x.base + b*batch_stride[0], dy.base +
 *                b*batch_stride[1], p[0] + b*batch_stride[2] (elements) -- the 64 transform positions of the
 *                Winograd-domain weight gradient of a 5x5 conv:
 *   WINO_DY      dM = A dY A^T per 4x4 tile of the output gradient: x = dy view, y = dM as [64][tiles][c] per sample,
 *                p[0] = A^T [4][8], kh x kw = tile grid
 *   WINO_DW      grad g [cout][25][cin] (p[1]) += G^T dU G of dU [64][cout][cin] (p[0]), p[2] = G [8][5]; cout, cin_g
 *   PACK_MULTI   every mode 0 / 1 / 2 PACK_W of a step in one launch: p[0] = dev table of hvn_pack_desc [cout], p[1] = dev int32
 *                [cout + 1] first workgroup (of 256 outputs) of each packing, batch_stride[0] = workgroups in total.  The table lives
 *                in device memory, so PACK_W's shape checks (cin % 32 / cout % 32, lead_pad >= rows, non-null pointers) cannot be
 *                made at the call: the CALLER makes them when it builds the table (train_engine._pack_ops raises on a bad entry)
 *   SPLIT_X3     p[0] = fp32 weight packings (batch_stride[0] granules of 32 floats, any concatenation of PACK_W outputs), p[1] = their
 *                three bf16 planes, [3][32] bf16 per granule: what a CONV with act_dtype 2 | 3 reads (the weights of a training step
 *                change every step, so the planes are made on the device after the PACK_W ops)
 */
enum { HVN_T_NET = 1, HVN_T_PACK_W = 2, HVN_T_BN_FWD = 3, HVN_T_BN_BWD = 4, HVN_T_WGRAD = 5, HVN_T_CONV0_WGRAD = 6,
       HVN_T_UPADD_BWD = 7, HVN_T_HEAD_BWD = 8, HVN_T_WINO_DY = 9, HVN_T_WINO_DW = 10, HVN_T_SPLIT_X3 = 11, HVN_T_PACK_MULTI = 12 };
typedef struct hvn_pack_desc {   /* one entry of PACK_MULTI's table = the fields of a PACK_W op */
    const float *src;            /* parameter [cout][kh*kw][cin_g] (dev) */
    float *dst;                  /* packing (dev) */
    int32_t cout, cin_g, groups, taps, mode, lead_pad;
    const float *gmat;           /* unused by modes 0 / 1 / 2 */
} hvn_pack_desc;
typedef struct hvn_top {
    int32_t kind, kh, kw, stride, pad_t, pad_l, groups, cout, cin_g, mode, lead_pad, _pad;
    hvn_view x, y, dx, dy;
    void *p[6];          /* dev */
    float eps, momentum;
    const hvn_op *net;   /* host */
    int64_t batch_stride[3]; /* WGRAD with nbatch > 1 */
    int32_t nbatch, _pad2;
} hvn_top;
HVN_API int hvn_run_train_plan(const hvn_top *ops, int n_ops, int batch, void *stream);
/* The same list with DETERMINISTIC cross-workgroup sums (reference: torch.use_deterministic_algorithms for run_desc.py:84-88's
 * loss.backward()).  hvn_run_train_plan ends WGRAD's split of the pixel sum, CONV0_WGRAD and HEAD_BWD in fp32 atomics, whose order the
 * hardware scheduler picks: the same step run twice differs in the last bits of its weight gradients.  With a workspace every such
 * workgroup STORES its partial tile into its own copy and a second launch adds the copies in a fixed order: bit-identical gradients
 * run to run (and box to box for equal `mode` hints).  workspace: dev, 16-byte aligned, >= hvn_train_workspace_bytes(ops, n_ops,
 * batch) bytes (HVN_E_SIZE otherwise), shared by all ops of the list (they run in stream order); NULL = hvn_run_train_plan. */
HVN_API int hvn_run_train_plan_ws(const hvn_top *ops, int n_ops, int batch, void *stream, void *workspace, size_t workspace_bytes);
HVN_API size_t hvn_train_workspace_bytes(const hvn_top *ops, int n_ops, int batch);
HVN_API const char *hvn_train_last_error(void);

/* Losses of run_desc.py:40-82 with opt.py:47-51 weights (np: bce + dice, hv: mse + msge, tp: bce + dice).
 * logits_* / grad_*: dev float32 NCHW [n][c][h][w] (c = 2, 2, nr_types); true_np / true_tp: dev int32 [n][h][w];
 * true_hv: dev float32 [n][h][w][2]; sums: dev double[64], zero before hvn_loss_forward, which adds this rank's
 * partial sums ([0] bce_np [1] bce_tp [2] mse [3] msge numerator [4] focus sum; [8+c] [10+c] [12+c] dice np
 * inse/l/r; [16+c] [32+c] [48+c] dice tp); the caller may SUM-all-reduce it over ranks, sets total_pixels to the
 * pixel count of the whole batch (all ranks), and hvn_loss_backward writes the logit gradients of this rank's
 * pixels.  sobel_ws: dev float32 [n][h][w][2] scratch carried from forward to backward. */
typedef struct hvn_loss {
    const float *logits_np, *logits_hv, *logits_tp;
    const int32_t *true_np, *true_tp;
    const float *true_hv;
    float *grad_np, *grad_hv, *grad_tp;
    double *sums;
    float *sobel_ws;
    int32_t n, h, w, nr_types;   /* nr_types = 0: no tp branch */
    double total_pixels;
    /* loss weights of run_desc.py:66-82 (`loss += loss_weight * term_loss`, table opt.py:47-51): np bce, np dice, hv mse,
     * hv msge, tp bce, tp dice; 0 = term absent from the table.  They scale the logit gradients (hvn_loss_backward); the
     * partial sums of hvn_loss_forward are the unweighted terms, which is what the reference tracks per term. */
    float weight[6];
    /* deterministic sums: dev double [partials_cap >= hvn_loss_partials_count(n, h, w)]; every workgroup of hvn_loss_forward stores
     * its 64 sums there and a second launch adds them to `sums` in a fixed order.  NULL: double atomics into `sums` (their order, and
     * with it the 16th digit of the sums, varies run to run). */
    double *partials;
    int64_t partials_cap;
} hvn_loss;
HVN_API int64_t hvn_loss_partials_count(int n, int h, int w);
HVN_API int hvn_loss_forward(const hvn_loss *l, void *stream);
HVN_API int hvn_loss_backward(const hvn_loss *l, void *stream);

/* Training targets: models/hovernet/targets.py:100-116 gen_targets (gen_instance_hv_map :17-96 with fix_mirror_padding,
 * remove_small_objects(30) on the crop, the unclamped 2-px box widening and its skip-at-the-near-border consequence).
 * ann: dev int32 [n][h][w] instance ids (0 = background); hv_map: dev float32 [n][crop_h][crop_w][2] = (x, y) offsets in
 * [-1, 1]; np_map: dev int32 [n][crop_h][crop_w] in {0, 1}.  Bit-exact with the reference. */
HVN_API size_t hvn_gen_targets_workspace_bytes(int n, int h, int w);
HVN_API int hvn_gen_targets(const int32_t *ann, int n, int h, int w, int crop_h, int crop_w, float *hv_map, int32_t *np_map,
                            void *workspace, size_t workspace_bytes, void *stream);

/* Training-time augmentation of a RESIDENT patch set: dataloader/train_loader.py:76-199 (FileLoader.__getitem__ + __get_augmentation)
 * and the image functions of dataloader/augs.py:36-113.  One record per OUTPUT sample, drawn on the host:
 *   shape part (image + annotation): out(y, x) = in[src](round_half_up(inv * (x', y', 1))) or 0 outside, where (x', y') is the
 *   output pixel after the flips, offset to the centre-crop window of the source patch (CropToFixedSize / cropping_center);
 *   input part (image only): kind 0 = cv2.GaussianBlur((p0, p1), 0) | 1 = cv2.medianBlur(p0) | 2 = additive Gaussian noise
 *   (noise_scale * z, z from the caller's N(0,1) buffer, one plane if !per_channel) | 3 = none; then order[0..3] in turn:
 *   0 = add_to_hue(hue), 1 = add_to_saturation(sat = 1 + draw), 2 = add_to_brightness(bright), 3 = add_to_contrast (returns its
 *   input unchanged in the reference, augs.py:96-97), < 0 = skip. */
typedef struct hvn_aug_sample {
    double  inv[6];            /* destination -> source affine: x_s = inv[0] x + inv[1] y + inv[2], y_s = inv[3] x + inv[4] y + inv[5] */
    int32_t src;               /* index of the source patch in the resident set */
    int32_t flip_lr, flip_ud;
    int32_t kind, p0, p1;
    int32_t per_channel;
    float   noise_scale;
    int32_t order[4];
    double  hue, sat, bright, contrast;
} hvn_aug_sample;
/* img: dev uint8 [P][h][w][3], ann: dev int32 [P][h][w][c] (c = 1 or 2: instance ids, types); out_img: dev uint8 [n][oh][ow][3],
 * out_ann: dev int32 [n][oh][ow][c]; prm: dev hvn_aug_sample[n] with 0 <= src < P. */
HVN_API int hvn_augment_shape(const uint8_t *img, const int32_t *ann, int n_resident, int h, int w, int c, const hvn_aug_sample *prm, int n,
                              int out_h, int out_w, uint8_t *out_img, int32_t *out_ann, void *stream);
/* src / dst: dev uint8 [n][h][w][3] (must not alias: the blurs read neighbours); noise: dev float32 [n][h][w][3] standard normal samples
 * (may be NULL when no record has kind 2). */
HVN_API int hvn_augment_input(const uint8_t *src, const hvn_aug_sample *prm, const float *noise, int n, int h, int w, uint8_t *dst, void *stream);

/* torch.optim.Adam (opt.py:38-44: lr 1e-4, betas (0.9, 0.999), eps 1e-8, no weight decay) over flat dev slabs;
 * step = 1 for the first update. */
HVN_API int hvn_adam_step(float *w, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2,
                          float eps, int step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HVN_H */
