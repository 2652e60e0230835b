// hvn_api.hip -- the C ABI (include/hvn.h) over the kernels: descriptor validation,
// plan execution, error text, optional per-launch timing of the conv kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/hvn.h"
#include "hvn_kernels.h"

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *a = "", long b = 0)
{
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

extern "C" {

int hvn_version(void) { return 104; }   // 1.04: + CONV act_dtype 2 | 3 (bf16x3 products), PACK_MULTI / SPLIT_X3 training ops, wsi_merge counters[5]; 1.02: + CHAIN op

#ifndef HVN_BUILD_ID
#define HVN_BUILD_ID "unstamped"
#endif
static const char hvn_build_tag[] = "hvn-build-id:" HVN_BUILD_ID;     // findable in the file's bytes without loading it
const char *hvn_build_id(void) { return hvn_build_tag + 13; }   // hash of the sources this binary was compiled from (hover_net_amd/lib.py:source_id)

const char *hvn_last_error(void) { return g_err; }

int hvn_device_ok(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ---- profiling of CONV launches (bench.py's roofline leg) -------------------------------
static bool g_prof = false;
static std::vector<hipEvent_t> g_ev;   // pairs
static size_t g_ev_used = 0;

int hvn_profile_enable(int on)
{
    g_prof = on != 0;
    g_ev_used = 0;
    return 0;
}

static void prof_mark(hipStream_t s)
{
    if (g_ev_used == g_ev.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        g_ev.push_back(e);
    }
    hipEventRecord(g_ev[g_ev_used++], s);
}

double hvn_profile_conv_ms(void)
{
    if (g_ev_used < 2) return -1.0;
    hipEventSynchronize(g_ev[g_ev_used - 1]);
    double total = 0.0;
    for (size_t i = 0; i + 1 < g_ev_used; i += 2) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, g_ev[i], g_ev[i + 1]);
        total += ms;
    }
    return total;
}

int hvn_profile_conv_launches(void) { return (int)(g_ev_used / 2); }

int hvn_profile_conv_ms_list(double *out, int cap)
{
    if (g_ev_used < 2) return 0;
    hipEventSynchronize(g_ev[g_ev_used - 1]);
    int n = 0;
    for (size_t i = 0; i + 1 < g_ev_used && n < cap; i += 2, ++n) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, g_ev[i], g_ev[i + 1]);
        out[n] = ms;
    }
    return n;
}

// ---- one op ---------------------------------------------------------------------------------
static bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

static int run_one(const hvn_op *op, int batch, hipStream_t s)
{
    switch (op->kind) {
    case HVN_OP_CONV0: {
        Conv0Args a;
        a.img = op->x.base;
        a.isn = op->x.sn; a.isy = op->x.sy; a.isx = op->x.sx; a.isc = op->x.sc ? op->x.sc : 1;
        a.is_f32 = op->x_dtype;
        a.H = op->x.h; a.W = op->x.w;
        a.w = op->w; a.bias = op->bias;
        a.y = (float *)op->y.base;
        a.ysn = op->y.sn; a.ysy = op->y.sy; a.ysx = op->y.sx;
        a.N = batch; a.Ho = op->y.h; a.Wo = op->y.w; a.pad = op->pad_t; a.relu = op->relu; a.out_bf16 = op->act_dtype == 1;
        if (op->kh != 7 || op->kw != 7 || op->x.c != 3 || op->y.c != 64 || !op->w || !op->bias)
            return fail(HVN_E_ARG, "conv0: expects 7x7x3->64 with bias%s", "");
        if (!aligned16(a.y) || (a.ysx & 3) || (a.ysy & 3) || (a.ysn & 3)) return fail(HVN_E_ARG, "conv0: output view not 16-byte aligned%s", "");
        return hvn_launch_conv0(a, s);
    }
    case HVN_OP_CONV: {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.x = (const float *)op->x.base;
        a.xsn = op->x.sn; a.xsy = op->x.sy; a.xsx = op->x.sx;
        a.H = op->x.h; a.W = op->x.w; a.Cin = op->x.c;
        a.w = op->w; a.bias = op->bias;
        a.pre_s = op->pre_scale; a.pre_b = op->pre_shift;
        a.post_s = op->post_scale; a.post_b = op->post_shift;
        a.res = (const float *)op->res.base;
        a.rsn = op->res.sn; a.rsy = op->res.sy; a.rsx = op->res.sx;
        a.y = (float *)op->y.base;
        a.ysn = op->y.sn; a.ysy = op->y.sy; a.ysx = op->y.sx;
        a.N = batch; a.Ho = op->y.h; a.Wo = op->y.w; a.Cout = op->cout;
        a.KH = op->kh; a.KW = op->kw; a.stride = op->stride; a.pad_t = op->pad_t; a.pad_l = op->pad_l;
        a.relu = op->relu;
        a.groups = op->groups > 1 ? op->groups : 1;
        a.x2 = (const float *)op->x2.base;
        a.x2sn = op->x2.sn; a.x2sy = op->x2.sy; a.x2sx = op->x2.sx;
        a.Cin2 = a.x2 ? op->x2.c : 0;
        a.stride2 = op->_rsv > 0 ? op->_rsv : 1;
        if (a.x2 && (op->kh != 1 || op->kw != 1 || a.stride != 1 || a.Cin2 % 32 || !aligned16(a.x2) || ((a.x2sn | a.x2sy | a.x2sx) & 3) ||
                     a.pre_s || (long)(a.Ho - 1) * a.stride2 >= op->x2.h || (long)(a.Wo - 1) * a.stride2 >= op->x2.w))
            return fail(HVN_E_ARG, "conv: second input needs a 1x1 stride-1 op without prologue and a view that covers the output grid%s", "");
        a.M = (long)batch * a.Ho * a.Wo;
        {   // EXPERIMENT, timing only (results are garbage): address dense 1x1 operands channel-BLOCKED, [C/32][H][W][32], inside the same buffers.
            // HVN_EXP_BLOCKED bit 0 = input, bit 1 = output + residual.  profiles/r02_experiments.md section 9.
            static int blk = -1, bsz = 32;
            if (blk < 0) {
                const char *e = getenv("HVN_EXP_BLOCKED");
                blk = e ? atoi(e) : 0;
                const char *b = getenv("HVN_EXP_BLOCK");     // channels per block: 32 (default) or 128
                bsz = b ? atoi(b) : 32;
            }
            const bool plain = op->kh == 1 && op->kw == 1 && a.stride == 1 && !a.x2 && op->nbatch <= 1 && op->act_dtype == 0;
            if (blk && plain && (bsz == 32 || bsz == 128)) {
                bool any = false;
                if ((blk & 1) && a.xsx == a.Cin && a.xsy == (long)a.W * a.Cin && a.xsn >= (long)a.H * a.W * a.Cin && a.H == a.Ho && a.W == a.Wo && a.Cin % bsz == 0) {
                    a.xsb = (long)a.H * a.W * bsz; a.xsx = bsz; a.xsy = (long)a.W * bsz;
                    any = true;
                }
                if ((blk & 2) && a.ysx == a.Cout && a.ysy == (long)a.Wo * a.Cout && a.ysn >= (long)a.Ho * a.Wo * a.Cout && a.Cout % bsz == 0) {
                    a.ysb = (long)a.Ho * a.Wo * bsz; a.ysx = bsz; a.ysy = (long)a.Wo * bsz;
                    any = true;
                    if (a.res && a.rsx == a.Cout && a.rsy == (long)a.Wo * a.Cout && a.rsn >= (long)a.Ho * a.Wo * a.Cout) {
                        a.rsb = a.ysb; a.rsx = bsz; a.rsy = a.ysy;
                    }
                }
                if (any) a.blk_shift = bsz == 32 ? 5 : 7;
            }
        }
        if (!a.x || !a.w || !a.y) return fail(HVN_E_ARG, "conv: null pointer%s", "");
        if (a.Cin % 32) return fail(HVN_E_ARG, "conv: input channels must be a multiple of 32 (got %s%ld)", "", a.Cin);
        if (!aligned16(a.x) || !aligned16(a.w) || (a.xsx & 3) || (a.xsy & 3) || (a.xsn & 3))
            return fail(HVN_E_ARG, "conv: input view / weights not 16-byte aligned%s", "");
        if ((a.pre_s && !aligned16(a.pre_s)) || (a.pre_b && !aligned16(a.pre_b)) || (!a.pre_s != !a.pre_b))
            return fail(HVN_E_ARG, "conv: prologue vectors must be 16-byte aligned and come in pairs%s", "");
        if (!a.post_s != !a.post_b) return fail(HVN_E_ARG, "conv: epilogue affine must come in pairs%s", "");
        if (op->y.c != op->cout) return fail(HVN_E_ARG, "conv: output view channels != cout%s", "");
        a.nbatch = op->nbatch > 1 ? op->nbatch : 1;
        a.xb = op->batch_stride[0]; a.wb = op->batch_stride[1]; a.yb = op->batch_stride[2];
        if (a.nbatch > 1 && ((a.xb | a.wb | a.yb) & 3)) return fail(HVN_E_ARG, "conv: batch strides must keep 16-byte alignment%s", "");
        const bool bf16 = op->act_dtype == 1;
        if (bf16 && (((a.xsx | a.xsy | a.xsn) & 7) || (a.x2 && ((a.x2sx | a.x2sy | a.x2sn) & 7)) || a.nbatch > 1))
            return fail(HVN_E_ARG, "conv(bf16): input strides must be multiples of 8 elements; no batched launch%s", "");
        if (g_prof) prof_mark(s);
        const bool x3 = op->act_dtype == 2 || op->act_dtype == 3;
        if (x3 && (a.groups > 1 || (a.nbatch > 1 && (a.wb & 7)))) return fail(HVN_E_ARG, "conv(bf16x3): no grouped convs; batch stride of the planes must keep 16-byte alignment%s", "");
        int rc;
        if (bf16 && (op->tile_n == 896 || op->tile_n == 640))            // LDS-DMA form of the bf16 convolution (hvn_conv_bf16g.hip)
            rc = hvn_launch_conv_bf16g(a, op->tile_n == 896 ? 256 : 128, s);
        else if (bf16)
            rc = hvn_launch_conv_bf16(a, op->tile_n, s);
        else if (x3 && (op->tile_n == 896 || op->tile_n == 640))      // LDS-DMA form, 256 | 128 pixels x 128 channels (hvn_conv_x3g.hip)
            rc = hvn_launch_conv_x3g(a, op->tile_n == 896 ? 256 : 128, op->act_dtype == 3 ? 6 : 9, s);
        else
            rc = x3 ? hvn_launch_conv_x3(a, op->tile_n, op->act_dtype == 3 ? 6 : 9, s) : hvn_launch_conv(a, op->tile_n, s);
        if (g_prof) prof_mark(s);
        if (rc) return fail(rc == -1 ? HVN_E_ARG : HVN_E_LAUNCH, "conv: launch failed (tile_n=%s%ld)", "", op->tile_n);
        return 0;
    }
    case HVN_OP_CHAIN: {
        ChainArgs a;
        memset(&a, 0, sizeof(a));
        a.x = (const float *)op->x.base;
        a.xsn = op->x.sn; a.xsy = op->x.sy; a.xsx = op->x.sx; a.K1 = op->x.c;
        a.x2 = (const float *)op->x2.base;
        a.x2sn = op->x2.sn; a.x2sy = op->x2.sy; a.x2sx = op->x2.sx;
        a.K1b = a.x2 ? op->x2.c : 0;
        a.stride2 = op->_rsv > 0 ? op->_rsv : 1;
        a.w1 = op->w;
        a.res = (const float *)op->res.base;
        a.rsn = op->res.sn; a.rsy = op->res.sy; a.rsx = op->res.sx;
        a.y = (float *)op->y.base;
        a.ysn = op->y.sn; a.ysy = op->y.sy; a.ysx = op->y.sx; a.C = op->cout;
        a.post_s = op->post_scale; a.post_b = op->post_shift;
        a.pre_s = op->pre_scale; a.pre_b = op->pre_shift;
        a.w2 = op->w2; a.bias2 = op->bias2; a.relu2 = 1;
        a.y2 = (float *)op->y2.base;
        a.y2sn = op->y2.sn; a.y2sy = op->y2.sy; a.y2sx = op->y2.sx; a.N2 = op->cout2;
        a.N = batch; a.Ho = op->y.h; a.Wo = op->y.w;
        a.M = (long)batch * a.Ho * a.Wo;
        a.bm = op->tile_n == 64 ? 64 : 128;      // CHAIN: tile_n = pixels per workgroup (0 / 128: 128)
        const bool cx3 = op->act_dtype == 2 || op->act_dtype == 3;       // products on the bf16 pipe from bf16x3 splits: w / w2 = plane packings
        const bool cbf = op->act_dtype == 1;                             // bf16 activations and weights (hvn_conv_chain_bf16.hip)
        if (op->act_dtype != 0 && !cx3 && !cbf) return fail(HVN_E_ARG, "chain: act_dtype 0 (fp32), 1 (bf16) or 2 | 3 (fp32 with bf16x3 products)%s", "");
        if (cbf) {
            if (!a.x || !a.w1 || !a.y || !a.w2 || !a.y2) return fail(HVN_E_ARG, "chain: null pointer%s", "");
            if (op->kh != 1 || op->kw != 1 || op->stride != 1 || op->pad_t || op->pad_l || op->relu || op->bias)
                return fail(HVN_E_ARG, "chain: the first conv is a plain 1x1 (no bias / relu of its own)%s", "");
            if (!hvn_chain_bf16_supported(a.K1, a.K1b, a.C, a.N2) || op->y.c != a.C || op->y2.c != a.N2)
                return fail(HVN_E_ARG, "chain (bf16): needs input channels in slabs of 64 (64 or 128 in all), cout %% 256 == 0, cout2 in {64, 128} (cout2 = %s%ld)", "", a.N2);
            if (op->x.h != a.Ho || op->x.w != a.Wo || op->y2.h != a.Ho || op->y2.w != a.Wo ||
                (a.x2 && ((long)(a.Ho - 1) * a.stride2 >= op->x2.h || (long)(a.Wo - 1) * a.stride2 >= op->x2.w)))
                return fail(HVN_E_ARG, "chain: views do not cover the output grid%s", "");
            if (a.res && (a.rsn != a.ysn || a.rsy != a.ysy || a.rsx != a.ysx || op->res.c != a.C))
                return fail(HVN_E_ARG, "chain: the residual view must have the output's strides%s", "");
            if (!aligned16(a.x) || !aligned16(a.w1) || !aligned16(a.w2) || !aligned16(a.y) || !aligned16(a.y2) || (a.res && !aligned16(a.res)) ||
                (a.x2 && !aligned16(a.x2)) || ((a.xsn | a.xsy | a.xsx | a.ysn | a.ysy | a.ysx | a.y2sn | a.y2sy | a.y2sx | a.x2sn | a.x2sy | a.x2sx) & 7))
                return fail(HVN_E_ARG, "chain (bf16): views / weights not 16-byte aligned%s", "");
            if ((!a.pre_s != !a.pre_b) || (!a.post_s != !a.post_b) || (a.pre_s && (!aligned16(a.pre_s) || !aligned16(a.pre_b))) ||
                (a.post_s && (!aligned16(a.post_s) || !aligned16(a.post_b))) || (a.bias2 && !aligned16(a.bias2)))
                return fail(HVN_E_ARG, "chain: per-channel vectors must be 16-byte aligned and come in pairs%s", "");
            if (g_prof) prof_mark(s);
            const int rcb = hvn_launch_conv_chain_bf16(a, s);
            if (g_prof) prof_mark(s);
            if (rcb) return fail(rcb == -1 ? HVN_E_ARG : HVN_E_LAUNCH, "chain (bf16): launch failed (cout2=%s%ld)", "", a.N2);
            return 0;
        }
        if (!a.x || !a.w1 || !a.y || !a.w2 || !a.y2) return fail(HVN_E_ARG, "chain: null pointer%s", "");
        if (op->kh != 1 || op->kw != 1 || op->stride != 1 || op->pad_t || op->pad_l || op->relu || op->bias)
            return fail(HVN_E_ARG, "chain: the first conv is a plain 1x1 (no bias / relu of its own)%s", "");
        if (!hvn_chain_supported(a.C, a.N2) || op->y.c != a.C || op->y2.c != a.N2 || a.K1 % 32 || a.K1b % 32 || a.K1 + a.K1b < 64)
            return fail(HVN_E_ARG, "chain: needs cout %% 64 == 0, cout2 in {64, 128}, input channels in slabs of 32 (>= 64) (cout2 = %s%ld)", "", a.N2);
        if (op->x.h != a.Ho || op->x.w != a.Wo || op->y2.h != a.Ho || op->y2.w != a.Wo ||
            (a.x2 && ((long)(a.Ho - 1) * a.stride2 >= op->x2.h || (long)(a.Wo - 1) * a.stride2 >= op->x2.w)))
            return fail(HVN_E_ARG, "chain: views do not cover the output grid%s", "");
        if (a.res && (a.rsn != a.ysn || a.rsy != a.ysy || a.rsx != a.ysx || op->res.c != a.C))
            return fail(HVN_E_ARG, "chain: the residual view must have the output's strides%s", "");
        if (!aligned16(a.x) || !aligned16(a.w1) || !aligned16(a.w2) || !aligned16(a.y) || !aligned16(a.y2) || (a.res && !aligned16(a.res)) ||
            (a.x2 && !aligned16(a.x2)) || ((a.xsn | a.xsy | a.xsx | a.ysn | a.ysy | a.ysx | a.y2sn | a.y2sy | a.y2sx | a.x2sn | a.x2sy | a.x2sx) & 3))
            return fail(HVN_E_ARG, "chain: views / weights not 16-byte aligned%s", "");
        if ((!a.pre_s != !a.pre_b) || (!a.post_s != !a.post_b) || (a.pre_s && (!aligned16(a.pre_s) || !aligned16(a.pre_b))) ||
            (a.post_s && (!aligned16(a.post_s) || !aligned16(a.post_b))) || (a.bias2 && !aligned16(a.bias2)))
            return fail(HVN_E_ARG, "chain: per-channel vectors must be 16-byte aligned and come in pairs%s", "");
        if (g_prof) prof_mark(s);
        int rc;
        if (cx3 && op->tile_n == 1152)           // input tile resident in registers, operands a chunk ahead (hvn_conv_chain_x3r.hip)
            rc = hvn_launch_conv_chain_x3r(a, op->act_dtype == 3 ? 6 : 9, s);
        else
            rc = cx3 ? hvn_launch_conv_chain_x3(a, op->act_dtype == 3 ? 6 : 9, s) : hvn_launch_conv_chain(a, s);
        if (g_prof) prof_mark(s);
        if (rc) return fail(rc == -1 ? HVN_E_ARG : HVN_E_LAUNCH, "chain: launch failed (cout2=%s%ld)", "", a.N2);
        return 0;
    }
    case HVN_OP_WINO_IN:
    case HVN_OP_WINO_OUT: {
        WinoArgs a;
        const bool in = op->kind == HVN_OP_WINO_IN;
        a.x = (const float *)op->x.base;
        a.xsn = op->x.sn; a.xsy = op->x.sy; a.xsx = op->x.sx;
        a.y = (float *)op->y.base;
        a.ysn = op->y.sn; a.ysy = op->y.sy; a.ysx = op->y.sx;
        a.mat = op->w; a.bias = op->bias;
        a.N = batch; a.H = in ? op->x.h : op->y.h; a.W = in ? op->x.w : op->y.w; a.C = in ? op->x.c : op->y.c;
        a.ty = op->kh; a.tx = op->kw; a.pad = op->pad_t; a.relu = op->relu;
        a.accum = (!in && op->res.base != nullptr) ? 1 : 0;
        a.lo = nullptr; a.lsn = a.lsy = a.lsx = 0;
        if (in && op->res.base) {       // WINO_IN with res: the input is nearest2x(res) + x, formed on the fly (UPADD fused into the transform)
            a.lo = (const float *)op->res.base;
            a.lsn = op->res.sn; a.lsy = op->res.sy; a.lsx = op->res.sx;
            if (op->res.h * 2 != op->x.h || op->res.w * 2 != op->x.w || op->res.c != op->x.c || !aligned16(a.lo) || ((a.lsn | a.lsy | a.lsx) & 3))
                return fail(HVN_E_ARG, "wino_in: the half-resolution input must be x.h/2 x x.w/2 x x.c and 16-byte aligned%s", "");
        }
        if (a.accum && op->res.base != op->y.base) return fail(HVN_E_ARG, "wino_out: res must alias y (accumulate in place)%s", "");
        const int n2 = in ? op->y.h : op->x.h;       // transform positions (m + r - 1)^2
        if (op->stride > 1 && op->_rsv > 1) {         // explicit F(m x m, r x r): stride = m, _rsv = r
            a.m = op->stride; a.r = op->_rsv;
        } else {                                     // legacy encoding by the number of positions: 5x5 filters
            a.m = n2 == 36 ? 2 : n2 == 64 ? 4 : n2 == 100 ? 6 : 0; a.r = 5;
        }
        if (!a.m || (a.m + a.r - 1) * (a.m + a.r - 1) != n2 || !((a.m == 2 && a.r == 5) || ((a.m == 4 || a.m == 6) && (a.r == 5 || a.r == 3))))
            return fail(HVN_E_ARG, "winograd transform: %s%ld transform positions do not match F(2,5) / F(4,5) / F(4,3) / F(6,3) / F(6,5)", "", (long)n2);
        if (!a.x || !a.y || !a.mat || a.ty <= 0 || a.tx <= 0 || (a.C & 3)) return fail(HVN_E_ARG, "winograd transform: bad descriptor%s", "");
        if (!aligned16(a.x) || !aligned16(a.y) || ((a.xsn | a.xsy | a.xsx | a.ysn | a.ysy | a.ysx) & 3))
            return fail(HVN_E_ARG, "winograd transform: views not 16-byte aligned%s", "");
        if (in && (op->y.w != a.ty * a.tx || op->y.c != a.C)) return fail(HVN_E_ARG, "wino_in: V must be [n*n][tiles][c]%s", "");
        if (!in && (op->x.w != a.ty * a.tx || op->y.h > a.m * a.ty || op->y.h <= a.m * (a.ty - 1) || op->y.w > a.m * a.tx ||
                    op->y.w <= a.m * (a.tx - 1) || op->x.c != a.C))
            return fail(HVN_E_ARG, "wino_out: M must be [n*n][tiles][cout] and the tiles must cover y%s", "");
        if (g_prof) prof_mark(s);
        int rc = in ? hvn_launch_wino_in(a, s) : hvn_launch_wino_out(a, s);
        if (g_prof) prof_mark(s);
        return rc;
    }
    case HVN_OP_UPADD: {
        UpAddArgs a;
        a.lo = (const float *)op->x.base;
        a.lsn = op->x.sn; a.lsy = op->x.sy; a.lsx = op->x.sx;
        a.skip = (const float *)op->res.base;
        a.ssn = op->res.sn; a.ssy = op->res.sy; a.ssx = op->res.sx;
        a.y = (float *)op->y.base;
        a.ysn = op->y.sn; a.ysy = op->y.sy; a.ysx = op->y.sx;
        a.N = batch; a.H = op->y.h; a.W = op->y.w; a.C = op->y.c; a.bf16 = op->act_dtype == 1;
        if (a.bf16 && ((a.lsx | a.lsy | a.lsn | a.ssx | a.ssy | a.ssn | a.ysx | a.ysy | a.ysn) & 7))
            return fail(HVN_E_ARG, "upadd(bf16): strides must be multiples of 8 elements%s", "");
        if (!a.lo || !a.skip || !a.y) return fail(HVN_E_ARG, "upadd: null pointer%s", "");
        if (op->x.h * 2 != a.H || op->x.w * 2 != a.W || op->res.h != a.H || op->res.w != a.W || op->x.c != a.C || op->res.c != a.C)
            return fail(HVN_E_ARG, "upadd: shape mismatch%s", "");
        if (!aligned16(a.lo) || !aligned16(a.skip) || !aligned16(a.y) || ((a.lsx | a.lsy | a.lsn | a.ssx | a.ssy | a.ssn | a.ysx | a.ysy | a.ysn) & 3))
            return fail(HVN_E_ARG, "upadd: views not 16-byte aligned%s", "");
        return hvn_launch_upadd(a, s);
    }
    case HVN_OP_HEAD: {
        HeadArgs a;
        a.x = (const float *)op->x.base;
        a.xsn = op->x.sn; a.xsy = op->x.sy; a.xsx = op->x.sx;
        a.w = op->w; a.bias = op->bias;
        a.y = (float *)op->y.base;
        a.N = batch; a.H = op->x.h; a.W = op->x.w; a.Cout = op->cout; a.in_bf16 = op->act_dtype == 1;
        if (a.in_bf16 && ((a.xsx | a.xsy | a.xsn) & 7)) return fail(HVN_E_ARG, "head(bf16): strides must be multiples of 8 elements%s", "");
        if (op->x.c != 64 || !a.w || !a.bias || !a.x || !a.y) return fail(HVN_E_ARG, "head: expects 64 input channels, weights and bias%s", "");
        if (!aligned16(a.x) || ((a.xsx | a.xsy | a.xsn) & 3)) return fail(HVN_E_ARG, "head: input view not 16-byte aligned%s", "");
        return hvn_launch_head(a, s);
    }
    case HVN_OP_PREDMAP: {
        PredMapArgs a;
        a.np = (const float *)op->x.base;
        a.hv = (const float *)op->res.base;
        a.tp = op->w;
        a.y = (float *)op->y.base;
        a.N = batch; a.H = op->y.h; a.W = op->y.w; a.nr_types = op->cout;
        if (!a.np || !a.hv || !a.y || (a.nr_types > 0 && !a.tp)) return fail(HVN_E_ARG, "predmap: null pointer%s", "");
        if (a.nr_types > 0 && !aligned16(a.y)) return fail(HVN_E_ARG, "predmap: output not 16-byte aligned%s", "");
        return hvn_launch_predmap(a, s);
    }
    default:
        return fail(HVN_E_ARG, "unknown op kind %s%ld", "", op->kind);
    }
}

}  // extern "C"
int hvn_internal_run_one(const hvn_op *op, int batch, hipStream_t s) { return run_one(op, batch, s); }
extern "C" {

int hvn_run_op(const hvn_op *op, int batch, void *stream)
{
    if (!op || batch <= 0) return fail(HVN_E_ARG, "run_op: bad arguments%s", "");
    int rc = run_one(op, batch, (hipStream_t)stream);
    if (rc == -2) return fail(HVN_E_LAUNCH, "launch failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

int hvn_extract_patches(const uint8_t *img, int h, int w, const int32_t *coords, int n_patches, int win, int pad_t, int pad_l,
                        uint8_t *out, void *stream)
{
    if (!img || !coords || !out || h <= 0 || w <= 0 || n_patches <= 0 || win <= 0 || pad_t < 0 || pad_l < 0)
        return fail(HVN_E_ARG, "extract_patches: bad arguments%s", "");
    int rc = hvn_launch_extract_patches(img, h, w, coords, n_patches, win, pad_t, pad_l, out, (hipStream_t)stream);
    if (rc) return fail(HVN_E_LAUNCH, "extract_patches: launch failed%s", "");
    return 0;
}

int hvn_run_plan(const hvn_op *ops, int n_ops, int batch, void *stream)
{
    if (!ops || n_ops <= 0 || batch <= 0) return fail(HVN_E_ARG, "run_plan: bad arguments%s", "");
    for (int i = 0; i < n_ops; ++i) {
        int rc = run_one(&ops[i], batch, (hipStream_t)stream);
        if (rc == -2) return fail(HVN_E_LAUNCH, "launch failed at op %s%ld", "", i);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
