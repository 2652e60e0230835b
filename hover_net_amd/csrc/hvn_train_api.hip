// hvn_train_api.hip -- C ABI of the training step (include/hvn.h, "training" section): descriptor validation and
// dispatch of hvn_top lists, the loss stages and the Adam update.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "../../include/hvn.h"
#include "hvn_kernels.h"

// PACK_MULTI's device table is written by the host as hvn_pack_desc (include/hvn.h) and read by the kernel as PackArgs
static_assert(sizeof(hvn_pack_desc) == sizeof(PackArgs) && offsetof(hvn_pack_desc, gmat) == offsetof(PackArgs, gmat) &&
              offsetof(hvn_pack_desc, lead_pad) == offsetof(PackArgs, lead_pad) && offsetof(hvn_pack_desc, cout) == offsetof(PackArgs, cout),
              "hvn_pack_desc must mirror PackArgs");

static thread_local char t_err[256] = "";
static int tfail(int code, const char *what, long i)
{
    snprintf(t_err, sizeof(t_err), "train op %ld: %s", i, what);
    return code;
}
static inline bool al16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
static inline bool view_ok(const hvn_view &v) { return v.base && al16(v.base) && ((v.sn | v.sy | v.sx) & 3) == 0 && (v.c & 3) == 0; }

// The arguments of the three op kinds that end in a cross-workgroup sum, filled from the descriptor (shared by the launch path and by
// hvn_train_workspace_bytes, so that both see the same launch shape).
static int wgrad_args(const hvn_top *t, int batch, WgradArgs &a)
{
    memset(&a, 0, sizeof(a));
    if (!view_ok(t->x) || !view_ok(t->dy) || !t->p[0]) return -1;
    a.x = (const float *)t->x.base; a.xsn = t->x.sn; a.xsy = t->x.sy; a.xsx = t->x.sx; a.H = t->x.h; a.W = t->x.w; a.Cin = t->x.c;
    a.dy = (const float *)t->dy.base; a.dsn = t->dy.sn; a.dsy = t->dy.sy; a.dsx = t->dy.sx; a.Ho = t->dy.h; a.Wo = t->dy.w; a.Cout = t->dy.c;
    a.dw = (float *)t->p[0];
    a.N = batch; a.KH = t->kh; a.KW = t->kw; a.stride = t->stride; a.pad_t = t->pad_t; a.pad_l = t->pad_l;
    a.groups = t->groups > 1 ? t->groups : 1; a.Cin_g = a.Cin / a.groups;
    a.nbatch = t->nbatch > 1 ? t->nbatch : 1;
    a.xb = t->batch_stride[0]; a.db = t->batch_stride[1]; a.wb = t->batch_stride[2];
    a.want_wgs = t->mode > 0 ? t->mode : 0;
    return 0;
}
static int conv0_wgrad_args(const hvn_top *t, int batch, Conv0WgradArgs &a)
{
    memset(&a, 0, sizeof(a));
    if (!t->x.base || !view_ok(t->dy) || !t->p[0] || t->dy.c != 64 || t->x.c != 3) return -1;
    a.img = (const uint8_t *)t->x.base; a.isn = t->x.sn; a.isy = t->x.sy; a.isx = t->x.sx; a.H = t->x.h; a.W = t->x.w;
    a.dy = (const float *)t->dy.base; a.ysn = t->dy.sn; a.ysy = t->dy.sy; a.ysx = t->dy.sx;
    a.dw = (float *)t->p[0];
    a.N = batch; a.Ho = t->dy.h; a.Wo = t->dy.w; a.pad = t->pad_t;
    return 0;
}
static int head_bwd_args(const hvn_top *t, int batch, HeadBwdArgs &a)
{
    memset(&a, 0, sizeof(a));
    if (!view_ok(t->x) || !view_ok(t->dx) || t->x.c != 64 || !t->p[0] || !t->p[1] || !t->p[2] || !t->p[3]) return -1;
    a.x = (const float *)t->x.base; a.xsn = t->x.sn; a.xsy = t->x.sy; a.xsx = t->x.sx;
    a.dx = (float *)t->dx.base; a.dsn = t->dx.sn; a.dsy = t->dx.sy; a.dsx = t->dx.sx;
    a.dl = (const float *)t->p[0]; a.w = (const float *)t->p[1]; a.dw = (float *)t->p[2]; a.db = (float *)t->p[3];
    a.N = batch; a.H = t->x.h; a.W = t->x.w; a.Cout = t->cout;
    return 0;
}

// ws / ws_floats: the deterministic-reduce workspace of hvn_run_train_plan_ws (NULL: atomics)
static int run_top(const hvn_top *t, int batch, hipStream_t s, long idx, float *ws, long ws_floats)
{
    switch (t->kind) {
    case HVN_T_NET:
        if (!t->net) return tfail(HVN_E_ARG, "null net op", idx);
        return hvn_internal_run_one(t->net, batch, s);
    case HVN_T_PACK_W: {
        PackArgs a;
        a.src = (const float *)t->p[0]; a.dst = (float *)t->p[1];
        a.cout = t->cout; a.cin_g = t->cin_g; a.groups = t->groups > 1 ? t->groups : 1; a.taps = t->kh * t->kw;
        a.mode = t->mode; a.lead_pad = t->lead_pad; a.gmat = (const float *)t->p[2];
        if (!a.src || !a.dst) return tfail(HVN_E_ARG, "pack: null pointer", idx);
        if ((a.mode == 3 || a.mode == 4) && (!a.gmat || a.taps != 25 || a.groups != 1 || (a.mode == 3 ? a.cin_g : a.cout) % 32))
            return tfail(HVN_E_ARG, "pack: Winograd transform needs a 5x5 ungrouped conv, G and k % 32 == 0", idx);
        if (a.mode == 0 && ((a.cin_g * a.groups) % 32 || a.lead_pad < a.cout)) return tfail(HVN_E_ARG, "pack: forward needs cin % 32 == 0", idx);
        if (a.mode == 1 && (a.cout % 32 || a.lead_pad < a.cin_g * a.groups)) return tfail(HVN_E_ARG, "pack: dgrad needs cout % 32 == 0", idx);
        return hvn_launch_pack_w(a, s);
    }
    case HVN_T_BN_FWD:
    case HVN_T_BN_BWD: {
        BnArgs a;
        memset(&a, 0, sizeof(a));
        const bool fwd = t->kind == HVN_T_BN_FWD;
        if (!view_ok(t->x) || !view_ok(t->y)) return tfail(HVN_E_ARG, "bn: bad z / a view", idx);
        a.z = (const float *)t->x.base; a.zsn = t->x.sn; a.zsy = t->x.sy; a.zsx = t->x.sx;
        a.a = (const float *)t->y.base; a.a_out = (float *)t->y.base; a.asn = t->y.sn; a.asy = t->y.sy; a.asx = t->y.sx;
        a.N = batch; a.H = t->x.h; a.W = t->x.w; a.C = t->x.c;
        a.ws = (double *)t->p[0]; a.save = (float *)t->p[1]; a.gamma = (const float *)t->p[2];
        a.eps = t->eps; a.momentum = t->momentum;
        if (!a.ws || !a.save || !a.gamma) return tfail(HVN_E_ARG, "bn: null pointer", idx);
        if (fwd) {
            a.beta = (const float *)t->p[3]; a.running_mean = (float *)t->p[4]; a.running_var = (float *)t->p[5];
            if (!a.beta || !a.running_mean || !a.running_var) return tfail(HVN_E_ARG, "bn forward: null pointer", idx);
            return hvn_launch_bn_forward(a, s);
        }
        if (!view_ok(t->dy)) return tfail(HVN_E_ARG, "bn backward: bad da view", idx);
        a.da = (const float *)t->dy.base; a.gsn = t->dy.sn; a.gsy = t->dy.sy; a.gsx = t->dy.sx;
        if (t->dx.base) {
            if (!view_ok(t->dx)) return tfail(HVN_E_ARG, "bn backward: bad dz view", idx);
            a.dz = (float *)t->dx.base; a.dsn = t->dx.sn; a.dsy = t->dx.sy; a.dsx = t->dx.sx;
        }
        a.dz_store = t->mode & 1;
        a.dgamma = (float *)t->p[3]; a.dbeta = (float *)t->p[4]; a.coef = (float *)t->p[5];
        if (!a.dgamma || !a.dbeta || !a.coef) return tfail(HVN_E_ARG, "bn backward: null pointer", idx);
        return hvn_launch_bn_backward(a, s);
    }
    case HVN_T_WGRAD: {
        WgradArgs a;
        if (wgrad_args(t, batch, a)) return tfail(HVN_E_ARG, "wgrad: bad view", idx);
        if ((long)batch * a.Ho * a.Wo >= (1L << 31)) return tfail(HVN_E_ARG, "wgrad: too many rows", idx);
        const int x3 = t->_pad;     // 6 | 9: products on the bf16 pipe (bf16x3 splits of both operands) where the shape has that form
        if (x3 != 0 && x3 != 6 && x3 != 9) return tfail(HVN_E_ARG, "wgrad: _pad selects the bf16x3 form with 6 or 9 partial products (0: fp32 pipe)", idx);
        a.part = ws; a.part_cap = ws_floats;
        int rc = (x3 && hvn_wgrad_x3_supported(a)) ? hvn_launch_wgrad_x3(a, x3, s) : hvn_launch_wgrad(a, s);
        if (rc == -1) return tfail(HVN_E_ARG, "wgrad: unsupported channel counts", idx);
        if (rc == -4) return tfail(HVN_E_SIZE, "wgrad: workspace smaller than hvn_train_workspace_bytes", idx);
        return rc;
    }
    case HVN_T_CONV0_WGRAD: {
        Conv0WgradArgs a;
        if (conv0_wgrad_args(t, batch, a)) return tfail(HVN_E_ARG, "conv0 wgrad: bad arguments", idx);
        a.part = ws; a.part_cap = ws_floats;
        int rc = hvn_launch_conv0_wgrad(a, s);
        if (rc == -4) return tfail(HVN_E_SIZE, "conv0 wgrad: workspace smaller than hvn_train_workspace_bytes", idx);
        return rc;
    }
    case HVN_T_UPADD_BWD: {
        UpAddBwdArgs a;
        memset(&a, 0, sizeof(a));
        if (!view_ok(t->dy)) return tfail(HVN_E_ARG, "upadd backward: bad dy view", idx);
        a.dy = (const float *)t->dy.base; a.ysn = t->dy.sn; a.ysy = t->dy.sy; a.ysx = t->dy.sx;
        if (t->dx.base) {
            if (!view_ok(t->dx)) return tfail(HVN_E_ARG, "upadd backward: bad dlo view", idx);
            a.dlo = (float *)t->dx.base; a.lsn = t->dx.sn; a.lsy = t->dx.sy; a.lsx = t->dx.sx;
        }
        if (t->y.base) {
            if (!view_ok(t->y)) return tfail(HVN_E_ARG, "upadd backward: bad dskip view", idx);
            a.dskip = (float *)t->y.base; a.ssn = t->y.sn; a.ssy = t->y.sy; a.ssx = t->y.sx;
        }
        a.N = batch; a.H = t->dy.h; a.W = t->dy.w; a.C = t->dy.c;
        int rc = hvn_launch_upadd_bwd(a, s);
        if (rc == -1) return tfail(HVN_E_ARG, "upadd backward: extent must be even", idx);
        return rc;
    }
    case HVN_T_HEAD_BWD: {
        HeadBwdArgs a;
        if (head_bwd_args(t, batch, a)) return tfail(HVN_E_ARG, "head backward: bad arguments", idx);
        a.part = ws; a.part_cap = ws_floats;
        int rc = hvn_launch_head_bwd(a, s);
        if (rc == -1) return tfail(HVN_E_ARG, "head backward: 1..16 output channels", idx);
        if (rc == -4) return tfail(HVN_E_SIZE, "head backward: workspace smaller than hvn_train_workspace_bytes", idx);
        return rc;
    }
    case HVN_T_WINO_DY: {
        WinoArgs a;
        memset(&a, 0, sizeof(a));
        if (!view_ok(t->x) || !t->y.base || !al16(t->y.base) || !t->p[0]) return tfail(HVN_E_ARG, "wino_dy: bad arguments", idx);
        a.x = (const float *)t->x.base; a.xsn = t->x.sn; a.xsy = t->x.sy; a.xsx = t->x.sx;
        a.y = (float *)t->y.base; a.ysn = t->y.sn; a.ysy = t->y.sy; a.ysx = t->y.sx;
        a.mat = (const float *)t->p[0];
        a.N = batch; a.H = t->x.h; a.W = t->x.w; a.C = t->x.c; a.ty = t->kh; a.tx = t->kw; a.m = 4;
        if (t->y.h != 64 || t->y.w != a.ty * a.tx || t->y.c != a.C || 4 * a.ty < a.H || 4 * a.tx < a.W)
            return tfail(HVN_E_ARG, "wino_dy: dM must be [64][tiles][c] and the tiles must cover dy", idx);
        return hvn_launch_wino_dy(a, s);
    }
    case HVN_T_PACK_MULTI:
        if (!t->p[0] || !t->p[1] || t->cout <= 0 || t->batch_stride[0] <= 0) return tfail(HVN_E_ARG, "pack_multi: bad arguments", idx);
        if (hvn_launch_pack_w_multi((const PackArgs *)t->p[0], (const int *)t->p[1], t->cout, (long)t->batch_stride[0], s))
            return tfail(HVN_E_ARG, "pack_multi: launch refused", idx);
        return 0;
    case HVN_T_SPLIT_X3:
        if (!t->p[0] || !t->p[1] || t->batch_stride[0] <= 0) return tfail(HVN_E_ARG, "split_x3: bad arguments", idx);
        if (hvn_launch_split_x3((const float *)t->p[0], (uint16_t *)t->p[1], (long)t->batch_stride[0], s)) return tfail(HVN_E_ARG, "split_x3: launch refused", idx);
        return 0;
    case HVN_T_WINO_DW:
        if (!t->p[0] || !t->p[1] || !t->p[2] || t->cout <= 0 || t->cin_g <= 0) return tfail(HVN_E_ARG, "wino_dw: bad arguments", idx);
        return hvn_launch_wino_dw((const float *)t->p[0], (float *)t->p[1], (const float *)t->p[2], t->cout, t->cin_g, s);
    default:
        return tfail(HVN_E_ARG, "unknown kind", idx);
    }
}

extern "C" {

const char *hvn_train_last_error(void) { return t_err[0] ? t_err : hvn_last_error(); }

int hvn_run_train_plan_ws(const hvn_top *ops, int n_ops, int batch, void *stream, void *workspace, size_t workspace_bytes)
{
    t_err[0] = 0;
    if (!ops || n_ops <= 0 || batch <= 0) return tfail(HVN_E_ARG, "run_train_plan: bad arguments", -1);
    if (workspace && !al16(workspace)) return tfail(HVN_E_ARG, "run_train_plan: workspace must be 16-byte aligned", -1);
    for (int i = 0; i < n_ops; ++i) {
        int rc = run_top(&ops[i], batch, (hipStream_t)stream, i, (float *)workspace, workspace ? (long)(workspace_bytes / sizeof(float)) : 0);
        if (rc == -2) return tfail(HVN_E_LAUNCH, hipGetErrorString(hipGetLastError()), i);
        if (rc) {
            if (!t_err[0]) snprintf(t_err, sizeof(t_err), "train op %d: %s", i, hvn_last_error());
            return rc;
        }
    }
    return 0;
}

int hvn_run_train_plan(const hvn_top *ops, int n_ops, int batch, void *stream) { return hvn_run_train_plan_ws(ops, n_ops, batch, stream, NULL, 0); }

size_t hvn_train_workspace_bytes(const hvn_top *ops, int n_ops, int batch)
{
    long need = 0;
    if (!ops || batch <= 0) return 0;
    for (int i = 0; i < n_ops; ++i) {
        const hvn_top *t = &ops[i];
        long f = 0;
        if (t->kind == HVN_T_WGRAD) {
            WgradArgs a;
            if (!wgrad_args(t, batch, a)) f = hvn_wgrad_part_floats(a, t->_pad);
        } else if (t->kind == HVN_T_CONV0_WGRAD) {
            Conv0WgradArgs a;
            if (!conv0_wgrad_args(t, batch, a)) f = hvn_conv0_wgrad_part_floats(a);
        } else if (t->kind == HVN_T_HEAD_BWD) {
            HeadBwdArgs a;
            if (!head_bwd_args(t, batch, a) && a.Cout >= 1 && a.Cout <= 16) f = hvn_head_bwd_part_floats(a);
        }
        if (f > need) need = f;
    }
    return (size_t)need * sizeof(float);
}

static int loss_common(const hvn_loss *l, LossArgs &a)
{
    if (!l || !l->logits_np || !l->logits_hv || !l->true_np || !l->true_hv || !l->sums || !l->sobel_ws) return HVN_E_ARG;
    if (l->nr_types > 0 && (!l->logits_tp || !l->true_tp)) return HVN_E_ARG;
    if (l->nr_types < 0 || l->nr_types > 16 || l->n <= 0 || l->h <= 0 || l->w <= 0) return HVN_E_ARG;
    memset(&a, 0, sizeof(a));
    a.l_np = l->logits_np; a.l_hv = l->logits_hv; a.l_tp = l->logits_tp;
    a.t_np = l->true_np; a.t_tp = l->true_tp; a.t_hv = l->true_hv;
    a.d_np = l->grad_np; a.d_hv = l->grad_hv; a.d_tp = l->grad_tp;
    a.sums = l->sums; a.gws = l->sobel_ws;
    a.N = l->n; a.H = l->h; a.W = l->w; a.T = l->nr_types;
    a.m_total = l->total_pixels;
    for (int i = 0; i < 6; ++i) {
        if (!(l->weight[i] >= 0.f)) return HVN_E_ARG;   // also rejects NaN
        a.wt[i] = l->weight[i];
    }
    a.parts = l->partials;
    a.parts_cap = l->partials ? (long)l->partials_cap : 0;
    return 0;
}

int hvn_loss_forward(const hvn_loss *l, void *stream)
{
    LossArgs a;
    if (loss_common(l, a)) return tfail(HVN_E_ARG, "loss: bad descriptor", -1);
    int rc = hvn_launch_loss(a, 0, (hipStream_t)stream);
    if (rc == -4) return tfail(HVN_E_SIZE, "loss: partials smaller than hvn_loss_partials_count doubles", -1);
    return rc == -2 ? tfail(HVN_E_LAUNCH, "loss forward launch failed", -1) : rc;
}

int hvn_loss_backward(const hvn_loss *l, void *stream)
{
    LossArgs a;
    if (loss_common(l, a) || !l->grad_np || !l->grad_hv || (l->nr_types > 0 && !l->grad_tp) || !(l->total_pixels > 0))
        return tfail(HVN_E_ARG, "loss: bad descriptor", -1);
    int rc = hvn_launch_loss(a, 1, (hipStream_t)stream);
    return rc == -2 ? tfail(HVN_E_LAUNCH, "loss backward launch failed", -1) : rc;
}

int64_t hvn_loss_partials_count(int n, int h, int w) { return (n > 0 && h > 0 && w > 0) ? (int64_t)hvn_loss_part_doubles(n, h, w) : 0; }

size_t hvn_gen_targets_workspace_bytes(int n, int h, int w) { return (n > 0 && h > 0 && w > 0) ? hvn_targets_ws_bytes(n, h, w) : 0; }

int hvn_gen_targets(const int32_t *ann, int n, int h, int w, int crop_h, int crop_w, float *hv_map, int32_t *np_map, void *workspace,
                    size_t workspace_bytes, void *stream)
{
    if (!ann || !hv_map || !np_map || !workspace || n <= 0 || h <= 0 || w <= 0 || crop_h <= 0 || crop_w <= 0 || crop_h > h || crop_w > w ||
        (long)h * w >= (1L << 31))
        return tfail(HVN_E_ARG, "gen_targets: bad arguments", -1);
    int rc = hvn_launch_gen_targets(ann, n, h, w, crop_h, crop_w, hv_map, np_map, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc == -4) return tfail(HVN_E_SIZE, "gen_targets: workspace too small", -1);
    return rc == -2 ? tfail(HVN_E_LAUNCH, "gen_targets launch failed", -1) : rc;
}

int hvn_augment_shape(const uint8_t *img, const int32_t *ann, int n_resident, int h, int w, int c, const hvn_aug_sample *prm, int n, int out_h,
                      int out_w, uint8_t *out_img, int32_t *out_ann, void *stream)
{
    if (!img || !ann || !prm || !out_img || !out_ann || n_resident <= 0 || n <= 0 || h <= 0 || w <= 0 || c < 1 || c > 4 || out_h <= 0 || out_w <= 0 ||
        out_h > h || out_w > w || (long)n_resident * h * w >= (1L << 40))
        return tfail(HVN_E_ARG, "augment_shape: bad arguments", -1);
    int rc = hvn_launch_aug_shape(img, ann, h, w, c, prm, n, out_h, out_w, out_img, out_ann, (hipStream_t)stream);
    return rc ? tfail(HVN_E_LAUNCH, "augment_shape launch failed", -1) : 0;
}

int hvn_augment_input(const uint8_t *src, const hvn_aug_sample *prm, const float *noise, int n, int h, int w, uint8_t *dst, void *stream)
{
    if (!src || !prm || !dst || src == dst || n <= 0 || h <= 0 || w <= 0) return tfail(HVN_E_ARG, "augment_input: bad arguments", -1);
    int rc = hvn_launch_aug_input(src, prm, noise, n, h, w, dst, (hipStream_t)stream);
    return rc ? tfail(HVN_E_LAUNCH, "augment_input launch failed", -1) : 0;
}

int hvn_adam_step(float *w, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps, int step,
                  void *stream)
{
    if (!w || !g || !m || !v || n <= 0 || step < 1 || !al16(w) || !al16(g) || !al16(m) || !al16(v)) return tfail(HVN_E_ARG, "adam: bad arguments", -1);
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    int rc = hvn_launch_adam(w, g, m, v, (long)n, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), (hipStream_t)stream);
    return rc == -2 ? tfail(HVN_E_LAUNCH, "adam launch failed", -1) : rc;
}

}  // extern "C"
