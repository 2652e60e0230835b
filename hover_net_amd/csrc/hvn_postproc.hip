// hvn_postproc.hip -- HoVer-Net instance separation on the GPU, batched over tiles.
//
// Stands behind /root/reference/models/hovernet/post_proc.py:26-90 (__proc_np_hv) and the
// third-party arithmetic it calls (cv2.normalize / Sobel / GaussianBlur / morphologyEx,
// scipy.ndimage label / binary_fill_holes, skimage watershed); the operation ORDER of every
// floating-point stage is the one pinned by oracle/hvn_oracle.c, so results are bit-identical
// (labels included).  Compiled with -ffp-contract=off: the only fused multiply-adds are the
// explicit fma()/fmaf() calls.
//
// All stages are HBM/latency-bound byte movers over [n][H*W] planes; one launch handles all
// n tiles (blockIdx.y = tile).  Connected components use a lock-free union-find with
// atomicMin (root = smallest raster index, which is also what makes scipy's raster-order
// label numbering a prefix count of root flags).  The marker-controlled watershed is
// skimage's global priority flood, whose tie order depends on its binary-heap layout
// (SURVEY.md Appendix B): it is replayed exactly, one tile per workgroup, lane 0 driving the
// heap (in LDS when the tile's heap fits, else in HBM scratch); parallelism comes from the
// tiles of the batch.
#include <hip/hip_runtime.h>
#include <vector>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/hvn.h"
#include "hvn_kernels.h"

#define PP_T 256

// ---------------------------------------------------------------------------------------------
// ordered-integer encodings for atomic min/max of floats / doubles
__device__ __forceinline__ unsigned f2o(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ unsigned long long d2o(double d)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double o2d(unsigned long long u)
{
    return __longlong_as_double((long long)((u >> 63) ? (u & 0x7fffffffffffffffull) : ~u));
}

// cv::borderInterpolate(BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// OpenCV normalize(NORM_MINMAX, 0, 1, CV_32F) coefficients (oracle: hvn_o_norm_coeffs)
__device__ __forceinline__ void norm_coeffs(double smin, double smax, float *a, float *b)
{
    double scale = 1.0 * ((smax - smin > DBL_EPSILON) ? 1. / (smax - smin) : 0.);
    scale = (double)(float)scale;
    double shift = (double)((float)0.0 - (float)(smin * scale));
    *a = (float)scale;
    *b = (float)shift;
}

// per-tile reduction slots
struct TileStat {
    unsigned h_min, h_max, v_min, v_max;               // ordered-uint of float32
    unsigned long long sh_min, sh_max, sv_min, sv_max;  // ordered-u64 of float64 sobel
    int n_comp;      // mask components that hold at least one marker
    int heap_top;    // bump pointer into the HBM heap plane (oversized components)
    int tie;         // a component saw two equal-valued age-0 heap items: replay the tile globally
    int max_area;    // largest component bounding box of the tile
    int dbg[8];      // HVN_WS_STATS: components replayed by 0 small LDS window, 1 bitmap window, 2 HBM window; 3 handed to the one-lane
                     // heap by a mixed-label marker tie, 4 by a full frontier; 5 component heap replays, 6 whole-tile replays
};

struct PPBuf {
    int n, H, W, C, c0;
    long P;  // H*W
    const float *pred;
    int32_t *inst;
    // planes of n*P elements
    int32_t *blb, *par, *cnt, *mk, *par2, *lab, *broot, *bsz;
    float *hraw, *vraw;
    double *rowh, *rowv, *sobh, *sobv, *overall, *dist, *blur;
    uint8_t *m8a, *m8b;
    TileStat *stat;
    // watershed scratch
    unsigned long long *heap;  // 2 x u64 per item, n*P items
    int no_wave;               // HVN_WS_WAVE=0: every component on the one-lane binary heap (A/B and tests)
    int no_bitmap;             // HVN_WS_BITMAP=0: windows beyond the LDS keep keys + labels in HBM scratch (the round-2 form, A/B)
};

// ---------------------------------------------------------------------------------------------
// union-find on a plane (indices local to the tile)
__device__ __forceinline__ int uf_load(const int32_t *par, int i)
{
    return __hip_atomic_load(par + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(const int32_t *par, int i)
{
    int p;
    while ((p = uf_load(par, i)) != i) i = p;
    return i;
}
__device__ void uf_union(int32_t *par, int a, int b)
{
    for (;;) {
        a = uf_find(par, a);
        b = uf_find(par, b);
        if (a == b) return;
        if (a < b) {
            int t = a;
            a = b;
            b = t;
        }
        int old = atomicMin(par + a, b);  // attach the larger root under the smaller
        if (old == a) return;
        a = old;
    }
}

// K1: threshold, planar copies of h / v, min/max of h and v, union-find init
__global__ __launch_bounds__(PP_T) void pp_init(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    float h = 0.f, v = 0.f;
    const bool in = i < b.P;
    if (in) {
        const float *p = b.pred + ((long)n * b.P + i) * b.C + b.c0;
        const int fg = p[0] >= 0.5f;  // post_proc.py:43
        h = p[1];
        v = p[2];
        const long g = (long)n * b.P + i;
        b.blb[g] = fg;
        b.par[g] = fg ? (int)i : -1;
        b.cnt[g] = 0;
        b.hraw[g] = h;
        b.vraw[g] = v;
    }
    unsigned hmin = in ? f2o(h) : 0xffffffffu, hmax = in ? f2o(h) : 0u;
    unsigned vmin = in ? f2o(v) : 0xffffffffu, vmax = in ? f2o(v) : 0u;
    for (int o = 32; o > 0; o >>= 1) {
        hmin = min(hmin, (unsigned)__shfl_xor((int)hmin, o));
        hmax = max(hmax, (unsigned)__shfl_xor((int)hmax, o));
        vmin = min(vmin, (unsigned)__shfl_xor((int)vmin, o));
        vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, o));
    }
    if ((threadIdx.x & 63) == 0) {
        TileStat *s = b.stat + n;
        atomicMin(&s->h_min, hmin);
        atomicMax(&s->h_max, hmax);
        atomicMin(&s->v_min, vmin);
        atomicMax(&s->v_max, vmax);
    }
}

// generic CCL stages over a binary plane encoded in par (>=0 : foreground)
__global__ __launch_bounds__(PP_T) void pp_ccl_merge(int32_t *par_all, int H, int W, long P)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= P) return;
    int32_t *par = par_all + (long)n * P;
    if (uf_load(par, (int)i) < 0) return;
    const int y = (int)(i / W), x = (int)(i - (long)y * W);
    if (x > 0 && uf_load(par, (int)i - 1) >= 0) uf_union(par, (int)i, (int)i - 1);
    if (y > 0 && uf_load(par, (int)i - W) >= 0) uf_union(par, (int)i, (int)i - W);
}

__global__ __launch_bounds__(PP_T) void pp_ccl_flatten_count(int32_t *par_all, int32_t *cnt_all, long P)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= P) return;
    int32_t *par = par_all + (long)n * P;
    if (par[i] < 0) return;
    const int r = uf_find(par, (int)i);
    par[i] = r;  // racing writers all store a valid ancestor; the final state is the root
    if (cnt_all) atomicAdd(cnt_all + (long)n * P + r, 1);
}

// K4: remove_small_objects(min_size=10) on the blob mask (post_proc.py:46-47) + prepare nothing else
__global__ __launch_bounds__(PP_T) void pp_blb_filter(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const long g = (long)n * b.P + i;
    const int r = b.par[g];
    const int sz = r >= 0 ? b.cnt[(long)n * b.P + r] : 0;
    const int keep = sz >= 10;
    b.blb[g] = keep;
    b.broot[g] = keep ? r : -1;   // component id of the watershed mask = smallest raster index
    b.bsz[g] = keep ? sz : 0;
}

// K6: Sobel-21 row pass of both maps (h: derivative taps, v: smoothing taps) on the
// min-max-normalised inputs (post_proc.py:49-57), float64 accumulation in tap order
__constant__ double c_kd[21] = {-1, -18, -152, -798, -2907, -7752, -15504, -23256, -25194, -16796, 0,
                                16796, 25194, 23256, 15504, 7752, 2907, 798, 152, 18, 1};
__constant__ double c_ks[21] = {1, 20, 190, 1140, 4845, 15504, 38760, 77520, 125970, 167960, 184756,
                                167960, 125970, 77520, 38760, 15504, 4845, 1140, 190, 20, 1};

__global__ __launch_bounds__(PP_T) void pp_sobel_row(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const TileStat *s = b.stat + n;
    float ha, hb, va, vb;
    norm_coeffs((double)o2f(s->h_min), (double)o2f(s->h_max), &ha, &hb);
    norm_coeffs((double)o2f(s->v_min), (double)o2f(s->v_max), &va, &vb);
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    const float *hr = b.hraw + (long)n * b.P + (long)y * b.W;
    const float *vr = b.vraw + (long)n * b.P + (long)y * b.W;
    double sh = 0., sv = 0.;
#pragma unroll
    for (int k = 0; k < 21; ++k) {
        const int xx = reflect101(x - 10 + k, b.W);
        const double hn = (double)fmaf(hr[xx], ha, hb);
        const double vn = (double)fmaf(vr[xx], va, vb);
        if (k == 0) {
            sh = c_kd[0] * hn;
            sv = c_ks[0] * vn;
        } else {
            sh += c_kd[k] * hn;
            sv += c_ks[k] * vn;
        }
    }
    b.rowh[(long)n * b.P + i] = sh;
    b.rowv[(long)n * b.P + i] = sv;
}

// K7: column pass (h: symmetric smoothing, v: antisymmetric derivative) + min/max of both
__global__ __launch_bounds__(PP_T) void pp_sobel_col(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    const bool in = i < b.P;
    double sh = 0., sv = 0.;
    if (in) {
        const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
        const double *rh = b.rowh + (long)n * b.P + x;
        const double *rv = b.rowv + (long)n * b.P + x;
        sh = c_ks[10] * rh[(long)y * b.W] + 0.0;
        sv = 0.0;
#pragma unroll
        for (int k = 1; k <= 10; ++k) {
            const long ya = (long)reflect101(y + k, b.H) * b.W, yb = (long)reflect101(y - k, b.H) * b.W;
            sh = fma(c_ks[10 + k], rh[ya] + rh[yb], sh);
            sv = fma(c_kd[10 + k], rv[ya] - rv[yb], sv);
        }
        b.sobh[(long)n * b.P + i] = sh;
        b.sobv[(long)n * b.P + i] = sv;
    }
    unsigned long long hmin = in ? d2o(sh) : ~0ull, hmax = in ? d2o(sh) : 0ull;
    unsigned long long vmin = in ? d2o(sv) : ~0ull, vmax = in ? d2o(sv) : 0ull;
    for (int o = 32; o > 0; o >>= 1) {
        hmin = min(hmin, (unsigned long long)__shfl_xor((long long)hmin, o));
        hmax = max(hmax, (unsigned long long)__shfl_xor((long long)hmax, o));
        vmin = min(vmin, (unsigned long long)__shfl_xor((long long)vmin, o));
        vmax = max(vmax, (unsigned long long)__shfl_xor((long long)vmax, o));
    }
    if ((threadIdx.x & 63) == 0) {
        TileStat *s = b.stat + n;
        atomicMin(&s->sh_min, hmin);
        atomicMax(&s->sh_max, hmax);
        atomicMin(&s->sv_min, vmin);
        atomicMax(&s->sv_max, vmax);
    }
}

// K8: 1 - normalize(sobel) (64f->32f), overall, dist (post_proc.py:59-74)
__global__ __launch_bounds__(PP_T) void pp_combine(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const TileStat *s = b.stat + n;
    float ha, hb, va, vb;
    norm_coeffs(o2d(s->sh_min), o2d(s->sh_max), &ha, &hb);
    norm_coeffs(o2d(s->sv_min), o2d(s->sv_max), &va, &vb);
    const long g = (long)n * b.P + i;
    const float nh = (float)fma(b.sobh[g], (double)ha, (double)hb);
    const float nv = (float)fma(b.sobv[g], (double)va, (double)vb);
    const float a = 1.0f - nh, c = 1.0f - nv;
    const float m = a > c ? a : c;  // np.maximum
    const int blb = b.blb[g];
    double ov = (double)m - (double)(1 - blb);
    if (ov < 0) ov = 0;
    b.overall[g] = ov;
    b.dist[g] = (1.0 - ov) * (double)blb;
}

// K9/K10: GaussianBlur 3x3 sigma 0 in float64 (post_proc.py:76), then the marker mask
// (post_proc.py:78-81) and the union-find init of its BACKGROUND for binary_fill_holes
__global__ __launch_bounds__(PP_T) void pp_gauss_row(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    const double *S = b.dist + (long)n * b.P + (long)y * b.W;
    double s = 0.25 * S[reflect101(x - 1, b.W)];
    s += 0.5 * S[x];
    s += 0.25 * S[reflect101(x + 1, b.W)];
    b.rowh[(long)n * b.P + i] = s;
}

__global__ __launch_bounds__(PP_T) void pp_gauss_col_marker(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    const double *R = b.rowh + (long)n * b.P + x;
    double s = 0.5 * R[(long)y * b.W] + 0.0;
    const double a = R[(long)reflect101(y + 1, b.H) * b.W], c = R[(long)reflect101(y - 1, b.H) * b.W];
    s = fma(0.25, a + c, s);
    const long g = (long)n * b.P + i;
    b.blur[g] = -s;
    const int ovb = b.overall[g] >= 0.4;
    int m = b.blb[g] - ovb;
    m = m < 0 ? 0 : m;
    b.mk[g] = m;
    b.par2[g] = m ? -1 : (int)i;  // background pixels form the union-find
}

// K11: binary_fill_holes: background components that do not reach the border are holes
__global__ __launch_bounds__(PP_T) void pp_border_flag(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    if (y != 0 && x != 0 && y != b.H - 1 && x != b.W - 1) return;
    const int r = b.par2[(long)n * b.P + i];
    if (r >= 0) b.cnt[(long)n * b.P + r] = 1;  // cnt doubles as "component touches the border"
}

__global__ __launch_bounds__(PP_T) void pp_fill(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const long g = (long)n * b.P + i;
    const int r = b.par2[g];
    b.m8a[g] = (r < 0 || b.cnt[(long)n * b.P + r] == 0) ? 1 : 0;
}

// K12: MORPH_OPEN with the 5x5 ellipse 00100/11111x3/00100; pixels outside the image never
// win the min (erode) nor the max (dilate)
__device__ __forceinline__ bool ell5(int j, int i) { return (j != -2 && j != 2) || i == 0; }

__global__ __launch_bounds__(PP_T) void pp_erode(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    const uint8_t *s = b.m8a + (long)n * b.P;
    uint8_t m = 255;
    for (int j = -2; j <= 2; ++j)
        for (int k = -2; k <= 2; ++k) {
            if (!ell5(j, k)) continue;
            const int yy = y + j, xx = x + k;
            if (yy < 0 || yy >= b.H || xx < 0 || xx >= b.W) continue;
            const uint8_t v = s[(long)yy * b.W + xx];
            if (v < m) m = v;
        }
    b.m8b[(long)n * b.P + i] = m;
}

__global__ __launch_bounds__(PP_T) void pp_dilate_init(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    const uint8_t *s = b.m8b + (long)n * b.P;
    uint8_t m = 0;
    for (int j = -2; j <= 2; ++j)
        for (int k = -2; k <= 2; ++k) {
            if (!ell5(j, k)) continue;
            const int yy = y + j, xx = x + k;
            if (yy < 0 || yy >= b.H || xx < 0 || xx >= b.W) continue;
            const uint8_t v = s[(long)yy * b.W + xx];
            if (v > m) m = v;
        }
    const long g = (long)n * b.P + i;
    b.par[g] = m ? (int)i : -1;  // union-find over the opened marker (post_proc.py:85)
    b.cnt[g] = 0;
}

// K13: scipy.ndimage.label numbering = 1 + number of component roots earlier in raster order
// (root = first pixel of its component).  One workgroup per tile, chunked block scan.
__global__ __launch_bounds__(1024) void pp_rank_roots(PPBuf b)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t *par = b.par + (long)n * b.P;
    int32_t *lab = b.lab + (long)n * b.P;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (long base = 0; base < b.P; base += 1024) {
        const long i = base + tid;
        const int flag = (i < b.P && par[i] == (int)i) ? 1 : 0;
        int v = flag;
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
        if (lane == 63) wsum[wv] = v;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (i < b.P) lab[i] = flag ? off + v : 0;  // inclusive count = 1-based label at roots
        __syncthreads();
        if (tid == 1023) carry = off + v;
        __syncthreads();
    }
}

// marker labels after remove_small_objects(10) (post_proc.py:86); labels are NOT renumbered
__global__ __launch_bounds__(PP_T) void pp_marker_labels(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const long g = (long)n * b.P + i;
    const int r = b.par[g];
    int l = 0;
    if (r >= 0 && b.cnt[(long)n * b.P + r] >= 10) l = b.lab[(long)n * b.P + r];
    b.mk[g] = l;
}

// ---------------------------------------------------------------------------------------------
// K14: skimage.segmentation.watershed(dist, markers, mask=blb) replayed exactly.
// Per-component replay.  The flood never crosses a 4-connected component of the mask, and inside one
// component the (value, age) order of its own heap items is the same whether the other components'
// items are interleaved or not -- EXCEPT among age-0 items (the initial markers) of equal value, whose
// order is an artefact of the global heap layout.  So every component is replayed
// by its own workgroup, out of LDS, over its bounding box; a component that ever pops an age-0 item while
// an equal (value, age 0) item is the new top raises the tile's `tie` flag and the tile is redone as ONE
// window (= skimage's global run).  VAL / OUT index the window [bh][bw]; OUT is -1 outside the mask
// (component).  Heap items are 8 bytes (age << 32 | window index): the value is looked up in VAL, which
// keeps the whole state at 20 B per pixel so that an 80x80 tile (or a 7.6k-pixel blob) fits the 160 KB LDS.
struct HItem {  // heap item with its value inline: one dependent LDS read per heap level
    double v;
    unsigned long long ai;  // age << 32 | window index
};

// Returns the number of tie events seen (saturating at 2).  `swap_first_tie`: at the first tie event pop the
// OTHER tied item first -- used to prove a 2-way tie harmless (both orders give the same labels).
// HM = heap mode: 0 = 8-byte items (value looked up in VAL), 1 = 16-byte items with the value inline, 2 = inline items
// with the first `cap` slots (the top levels of the heap, where every pop's sift-down spends its time) in LDS
// (`heap_raw`) and the rest in HBM (`heap_far`, indexed by the same slot number): windows larger than the LDS.
template <int HM, typename VP, typename OP>
__device__ int ws_flood_window(void *heap_raw, VP val, OP out, int bh, int bw, bool swap_first_tie, HItem *heap_far = nullptr, int cap = 0)
{
    typedef unsigned long long u64;
    constexpr bool INLINE_VAL = HM != 0;
    HItem *h16 = (HItem *)heap_raw;
    u64 *h8 = (u64 *)heap_raw;
    const int A = bh * bw;
    int hn = 0;
    int ties = 0;
    auto less = [](double va, u64 a, double vb, u64 b) { return va != vb ? va < vb : (a >> 32) < (b >> 32); };
    auto get = [&](int i, u64 &it, double &v) {
        if constexpr (HM == 2) {
            const HItem t = i < cap ? h16[i] : heap_far[i];
            it = t.ai;
            v = t.v;
        } else if constexpr (INLINE_VAL) {
            const HItem t = h16[i];
            it = t.ai;
            v = t.v;
        } else {
            it = h8[i];
            v = val[(unsigned)it];
        }
    };
    auto put = [&](int i, u64 it, double v) {
        if constexpr (HM == 2) {
            if (i < cap) h16[i] = HItem{v, it};
            else heap_far[i] = HItem{v, it};
        } else if constexpr (INLINE_VAL) h16[i] = HItem{v, it};
        else h8[i] = it;
    };
    auto push = [&](u64 it, double itv) {
        int child = hn++;
        while (child > 0) {
            const int parent = (child + 1) / 2 - 1;
            u64 pi;
            double pv;
            get(parent, pi, pv);
            if (less(itv, it, pv, pi)) {
                put(child, pi, pv);
                child = parent;
            } else
                break;
        }
        put(child, it, itv);
    };
    for (int i = 0; i < A; ++i)
        if (out[i] > 0) push((u64)(unsigned)i, val[i]);
    unsigned age = 0;
    while (hn) {
        u64 top;
        double topv;
        get(0, top, topv);
        --hn;
        if (hn > 0) {
            u64 last;
            double lastv;
            get(hn, last, lastv);
            int i = 0;
            for (;;) {
                const int l = 2 * i + 1, r = l + 1;
                if (l >= hn) break;
                u64 li, ri = 0;
                double lv, rv = 0.;
                get(l, li, lv);
                if (r < hn) get(r, ri, rv);  // both children are fetched before either is compared
                int s = i;
                u64 si = last;
                double sv = lastv;
                if (less(lv, li, lastv, last)) {
                    s = l;
                    si = li;
                    sv = lv;
                }
                if (r < hn && less(rv, ri, sv, si)) {
                    s = r;
                    si = ri;
                    sv = rv;
                }
                if (s == i) break;
                put(i, si, sv);
                i = s;
            }
            put(i, last, lastv);
            if ((top >> 32) == 0) {
                u64 nt;
                double ntv;
                get(0, nt, ntv);
                if ((nt >> 32) == 0 && ntv == topv) {
                    if (ties == 0 && swap_first_tie) {  // equal keys: exchanging them keeps the heap valid
                        put(0, top, topv);
                        top = nt;
                    }
                    if (ties < 2) ++ties;
                }
            }
        }
        const int idx = (int)(unsigned)top;
        const int y = idx / bw, x = idx - y * bw;
        const int lab = out[idx];
        const int nb[4] = {idx - bw, idx - 1, idx + 1, idx + bw};  // neighbour order of skimage
        const bool ok[4] = {y > 0, x > 0, x < bw - 1, y < bh - 1};
        int oq[4];
        double nv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) oq[k] = ok[k] ? out[nb[k]] : -1;  // four independent reads in flight
#pragma unroll
        for (int k = 0; k < 4; ++k) nv[k] = (HM == 2 && ok[k]) ? val[nb[k]] : 0.;  // HBM values: fetched together, not one per push
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (oq[k] != 0) continue;  // -1: not in the mask / this component, >0: already labelled
            const int q = nb[k];
            age += 1;
            out[q] = lab;
            push(((u64)age << 32) | (unsigned)q, HM == 2 ? nv[k] : val[q]);
        }
    }
    return ties;
}

// ---- wave-cooperative replay (windows whose state fits the launch's LDS) --------------------------------------------------------
// skimage pops the heap item with the smallest (value, age).  Ages are unique (one global counter) except for the initial
// markers (all age 0), so -- marker ties aside -- WHICH priority queue is used cannot change the result: any structure that
// always extracts the exact minimum replays the same flood.  A binary heap walked by one lane costs ~8 dependent LDS round
// trips per pop; here the frontier is an UNSORTED array in LDS and all 64 lanes find its minimum together: every lane scans
// its slots (one 16-byte read for frontiers up to 64 items), the minimum of the sortable 64-bit value keys is reduced across
// the wave with DPP row shifts / row broadcasts (no LDS traffic), then the minimum age among the lanes holding that value.
// Push = append; pop = move the last item into the hole.  The four neighbours of the popped pixel are handled by four lanes at
// once (ages handed out in skimage's neighbour order by a ballot prefix count); the window carries a one-pixel border of
// "not in the mask" so no coordinate is ever range-checked.  A marker tie (two age-0 items of equal value at the minimum) has no
// defined order here: the component reports it and is redone on the one-lane binary heap (ws_window), which knows how to
// prove a two-way tie harmless or else hands the tile to the exact whole-tile replay.
__device__ inline unsigned long long ws_key(double v)
{
    v = v + 0.0;                                     // -0.0 -> +0.0: skimage compares doubles, for which they are equal
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);   // order-preserving map of finite doubles onto unsigned integers
}

template <int CTRL, int RMASK>
__device__ inline unsigned long long dpp_min_u64(unsigned long long x)
{
    const int lo = (int)(unsigned)x, hi = (int)(unsigned)(x >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, RMASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, RMASK, 0xf, false);
    const unsigned long long y = ((unsigned long long)(unsigned)hi2 << 32) | (unsigned)lo2;
    return y < x ? y : x;
}

template <int CTRL, int RMASK>
__device__ inline unsigned dpp_min_u32(unsigned x)
{
    const unsigned y = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, RMASK, 0xf, false);
    return y < x ? y : x;
}

// minimum over the 64 lanes, returned to every lane: row_shr 1/2/4/8 leave each row's minimum in its lane 15,
// row_bcast:15 / row_bcast:31 carry it to lane 63
__device__ inline unsigned long long wave_min_u64(unsigned long long x)
{
    x = dpp_min_u64<0x111, 0xf>(x);
    x = dpp_min_u64<0x112, 0xf>(x);
    x = dpp_min_u64<0x114, 0xf>(x);
    x = dpp_min_u64<0x118, 0xf>(x);
    x = dpp_min_u64<0x142, 0xa>(x);
    x = dpp_min_u64<0x143, 0xc>(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ inline unsigned wave_min_u32(unsigned x)
{
    x = dpp_min_u32<0x111, 0xf>(x);
    x = dpp_min_u32<0x112, 0xf>(x);
    x = dpp_min_u32<0x114, 0xf>(x);
    x = dpp_min_u32<0x118, 0xf>(x);
    x = dpp_min_u32<0x142, 0xa>(x);
    x = dpp_min_u32<0x143, 0xc>(x);
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// One component (root >= 0) over its bounding box, all 64 lanes.  `kv` / `out` hold the bordered window (A2 = (bh+2)(bw+2)
// entries: LDS for windows that fit, HBM scratch -- L2-resident -- for larger ones), `fk` / `fa` / `fi` the frontier (`cap`
// slots, always LDS).  Returns true when the component must be redone on the binary heap: a marker tie was met, or the
// frontier outgrew its slots.
template <bool BIG>
__device__ __forceinline__ bool ws_window_wave(PPBuf &b, int n, int root, int y0, int x0, int bh, int bw, unsigned long long *kv_far, int *out_far,
                                               int cap)
{
    typedef unsigned long long u64;
    // the launch's dynamic LDS, named here so that the compiler sees LDS addresses (ds_read / ds_write, not flat accesses)
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds_w[];
    const int lane = threadIdx.x;
    const long g0 = (long)n * b.P;
    const int bw2 = bw + 2, A2 = bw2 * (bh + 2);
    u64 *kv, *fk;
    unsigned *fa, *fi;
    int *out;
    if constexpr (BIG) {
        kv = kv_far;
        out = out_far;
        fk = (u64 *)ws_lds_w;
        fa = (unsigned *)(fk + cap);
        fi = fa + cap;
    } else {
        kv = (u64 *)ws_lds_w;
        fk = kv + A2;
        fa = (unsigned *)(fk + A2);
        fi = fa + A2;
        out = (int *)(fi + A2);
    }
    for (int t = lane; t < A2; t += 64) {
        const int ry = t / bw2;
        const int yy = ry - 1, xx = t - ry * bw2 - 1;
        int o = -1;
        u64 k = 0;
        if ((unsigned)yy < (unsigned)bh && (unsigned)xx < (unsigned)bw) {
            const long gi = g0 + (long)(y0 + yy) * b.W + (x0 + xx);
            if (b.broot[gi] == root) {
                o = b.mk[gi];
                k = ws_key(b.blur[gi]);
            }
        }
        out[t] = o;
        kv[t] = k;
    }
    __syncthreads();
    int nf = 0;
    for (int base = 0; base < A2; base += 64) {   // the markers, in any order: their keys decide
        const int i = base + lane;
        // only marker pixels with an unlabelled 4-neighbour enter the frontier: popping an interior marker pixel labels nothing,
        // pushes nothing and advances no age, so leaving it out is exact -- and keeps the frontier to the marker BORDERS
        // (a marker pixel is never on the window's border row / column, so i +- 1 and i +- bw2 stay inside the window)
        bool m = i < A2 && out[i] > 0;
        if (m) m = out[i - bw2] == 0 || out[i - 1] == 0 || out[i + 1] == 0 || out[i + bw2] == 0;
        const u64 mask = __ballot(m);
        if (nf + __popcll(mask) > cap) return true;   // (uniform) no room for the markers: binary-heap path
        if (m) {
            const int s = nf + __popcll(mask & ((1ull << lane) - 1ull));
            fk[s] = kv[i];
            fa[s] = 0u;
            fi[s] = (unsigned)i;
        }
        nf += __popcll(mask);
    }
    __syncthreads();
    unsigned age = 0;
    bool tie = false;
    while (nf > 0) {
        if (nf + 4 > cap) {                       // the frontier outgrew its LDS slots
            tie = true;
            break;
        }
        u64 bk = ~0ull;
        unsigned ba = 0xffffffffu;
        int bs = 0, cnt = 0;
        for (int s = lane; s < nf; s += 64) {
            const u64 k = fk[s];
            const unsigned a = fa[s];
            if (k < bk || (k == bk && a < ba)) {
                bk = k;
                ba = a;
                bs = s;
                cnt = 1;
            } else if (k == bk && a == ba)
                ++cnt;
        }
        const u64 gk = wave_min_u64(bk);
        const unsigned ga = wave_min_u32(bk == gk ? ba : 0xffffffffu);
        const bool mine = bk == gk && ba == ga && cnt > 0;
        const u64 win = __ballot(mine);
        if (ga == 0u && (__popcll(win) > 1 || __any(mine && cnt > 1))) {
            // Marker tie: several age-0 items share the minimal value.  Their order in the reference is an artefact of its
            // binary heap's layout -- but if they all carry the SAME label the order cannot change any label: until the next
            // foreign item pops, only items of that label pop and push (a contiguous block of ages), the set of pixels they
            // label is a reachability closure that does not depend on the order, and against every foreign item the block's
            // ages compare the same way whichever order was taken (checked on ~1900 tie-heavy random cases against the exact
            // heap model, tests/test_oracle_postproc.py).  Only a tie between DIFFERENT labels is ambiguous.
            unsigned lmin = 0xffffffffu, lmax = 0u;
            for (int s = lane; s < nf; s += 64)
                if (fk[s] == gk && fa[s] == 0u) {
                    const unsigned l = (unsigned)out[fi[s]];
                    lmin = l < lmin ? l : lmin;
                    lmax = l > lmax ? l : lmax;
                }
            if (wave_min_u32(lmin) != ~wave_min_u32(~lmax)) {
                if (lane == 0) atomicAdd(&b.stat[n].dbg[3], 1);
                tie = true;
                break;
            }
        }
        const int wl = __ffsll((long long)win) - 1;
        const int gs = __builtin_amdgcn_readlane(bs, wl);
        const int idx = (int)fi[gs];
        const int lab = out[idx];
        --nf;
        if (gs != nf && lane == 0) {   // the last item fills the hole
            fk[gs] = fk[nf];
            fa[gs] = fa[nf];
            fi[gs] = fi[nf];
        }
        bool unl = false;
        int q = 0;
        if (lane < 4) {                // skimage's neighbour order: up, left, right, down
            q = idx + (lane == 0 ? -bw2 : lane == 1 ? -1 : lane == 2 ? 1 : bw2);
            unl = out[q] == 0;
        }
        const u64 um = __ballot(unl);
        if (unl) {
            const int r = __popcll(um & ((1ull << lane) - 1ull));
            out[q] = lab;              // labelled at push time
            const int s = nf + r;
            fk[s] = kv[q];
            fa[s] = age + 1u + (unsigned)r;
            fi[s] = (unsigned)q;
        }
        const int np = __popcll(um);
        age += (unsigned)np;
        nf += np;
        __syncthreads();
    }
    __syncthreads();
    if (!tie)
        for (int t = lane; t < A2; t += 64) {
            const int v = out[t];
            if (v >= 0) {
                const int ry = t / bw2;
                b.inst[g0 + (long)(y0 + ry - 1) * b.W + (x0 + t - ry * bw2 - 1)] = v;
            }
        }
    __syncthreads();
    return tie;
}

// Windows beyond the LDS (a clump of many nuclei, a blob that fills the tile): the same wave-cooperative replay on a COMPACT
// state.  Per pixel the flood only ever asks "in the component and still unlabelled?" -- one bit -- so the window lives in LDS as
// a bitmap (a 164 x 164 tile: 3.4 KB; up to ~0.5 Mpixel windows fit), every frontier item carries its label (no label read per
// pop), labels are written straight to the instance map when a pixel is pushed (fire-and-forget stores), and the only HBM
// (L2-resident scratch) access a pop waits for is the value key of the up to four pixels it pushes.  Was: keys AND labels of the
// window in HBM scratch, three dependent L2 round trips per pop.
// Such windows have frontiers of thousands of items (every border pixel of hundreds of noise markers), too many to scan per pop:
// the frontier is a 64-ary TOURNAMENT -- slots in blocks of 64, each block's minimum (key, age, slot) cached; a pop reads the
// cached minima (one or two per lane), takes the wave minimum, re-derives the one block it emptied a slot of (64 lanes, one slot
// each), and a push compares its item with one cached minimum.  Freed slots are recycled through a stack, so nothing moves.
// Exactness is that of any exact-minimum queue (see ws_window_wave); for marker ties every block also caches how many of its
// items equal its minimum and their label range, so "one label or several?" never needs a scan either.
struct WsBlk {
    unsigned long long k;   // minimum value key of the block's live items (~0: block empty)
    unsigned a, s;          // its age, its slot
    unsigned c, l0, l1;     // items equal to the minimum (meaningful at age 0: marker ties), their smallest / largest label
    unsigned pad;
};

__device__ __forceinline__ bool ws_window_wave_bitmap(PPBuf &b, int n, int root, int y0, int x0, int bh, int bw, unsigned long long *kv, int lds_bytes)
{
    typedef unsigned long long u64;
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds_w[];
    const int lane = threadIdx.x;
    const long g0 = (long)n * b.P;
    const int bw2 = bw + 2, A2 = bw2 * (bh + 2);
    const int nw = (A2 + 63) >> 6;
    u64 *um = (u64 *)ws_lds_w;            // bit t: window pixel t belongs to the component and has no label yet
    u64 *mm = um + nw;                    // bit t: marker pixel (used while the frontier is seeded)
    constexpr int NPEND = 256;            // labels waiting to be written to the instance map (see flush below)
    const int cap = ((lds_bytes - 16 * nw - 8 * NPEND) / (24 * 64 + (int)sizeof(WsBlk))) * 64;     // slots, whole blocks
    unsigned *pend_i = (unsigned *)(mm + nw), *pend_l = pend_i + NPEND;
    WsBlk *blk = (WsBlk *)(pend_l + NPEND);
    u64 *fk = (u64 *)(blk + cap / 64);
    unsigned *fa = (unsigned *)(fk + cap), *fi = fa + cap, *fl = fi + cap, *fs = fl + cap;
    for (int base = 0; base < A2; base += 64) {
        const int t = base + lane;
        bool unl = false, mrk = false;
        if (t < A2) {
            const int ry = t / bw2;
            const int yy = ry - 1, xx = t - ry * bw2 - 1;
            if ((unsigned)yy < (unsigned)bh && (unsigned)xx < (unsigned)bw) {
                const long gi = g0 + (long)(y0 + yy) * b.W + (x0 + xx);
                u64 k = 0;
                if (b.broot[gi] == root) {
                    const int o = b.mk[gi];
                    k = ws_key(b.blur[gi]);
                    b.inst[gi] = o;       // marker label, or 0 until the flood reaches the pixel
                    unl = o == 0;
                    mrk = o > 0;
                }
                kv[yy * bw + xx] = k;     // keys of the bh x bw interior only: a window that IS the tile still fits the tile's scratch plane
            }
        }
        const u64 bu = __ballot(unl), bm = __ballot(mrk);
        if (lane == 0) {
            um[base >> 6] = bu;
            mm[base >> 6] = bm;
        }
    }
    for (int s = lane; s < cap; s += 64) {   // every slot starts dead
        fk[s] = ~0ull;
        fa[s] = ~0u;
    }
    __threadfence_block();
    __syncthreads();
    auto bit = [](const u64 *w, int t) { return (w[t >> 6] >> (t & 63)) & 1ull; };
    int top = 0;                          // slots ever used
    for (int base = 0; base < A2; base += 64) {   // marker pixels with an unlabelled neighbour (see ws_window_wave)
        const int t = base + lane;
        bool m = t < A2 && bit(mm, t);
        if (m) m = bit(um, t - bw2) || bit(um, t - 1) || bit(um, t + 1) || bit(um, t + bw2);
        const u64 mask = __ballot(m);
        if (top + __popcll(mask) + 4 > cap) {           // (uniform) no room: binary-heap path
            if (lane == 0) atomicAdd(&b.stat[n].dbg[4], 1);
            return true;
        }
        if (m) {
            const int s = top + __popcll(mask & ((1ull << lane) - 1ull));
            const int ry = t / bw2;
            fk[s] = kv[(ry - 1) * bw + (t - ry * bw2 - 1)];
            fa[s] = 0u;
            fi[s] = (unsigned)t;
            fl[s] = (unsigned)b.mk[g0 + (long)(y0 + ry - 1) * b.W + (x0 + t - ry * bw2 - 1)];
        }
        top += __popcll(mask);
    }
    __syncthreads();
    // the cached minimum of block j from its 64 slots, one per lane
    auto rebuild = [&](int j) {
        const int s = 64 * j + lane;
        const u64 k = fk[s];
        const unsigned a = fa[s];
        const u64 gk = wave_min_u64(k);
        const unsigned am = wave_min_u32(k == gk ? a : 0xffffffffu);
        const bool hit = k == gk && a == am && gk != ~0ull;
        const u64 hm = __ballot(hit);
        unsigned l0 = 0, l1 = 0;
        if (hm && am == 0u) {
            const unsigned l = hit ? fl[s] : 0u;
            l0 = wave_min_u32(hit ? l : 0xffffffffu);
            l1 = ~wave_min_u32(hit ? ~l : 0xffffffffu);
        }
        if (lane == 0) {
            WsBlk w;
            w.k = gk;
            w.a = am;
            w.s = hm ? (unsigned)(64 * j + __ffsll((long long)hm) - 1) : 0u;
            w.c = (unsigned)__popcll(hm);
            w.l0 = l0;
            w.l1 = l1;
            w.pad = 0;
            blk[j] = w;
        }
    };
    for (int j = 0; j < cap / 64; ++j) {
        if (64 * j < top) rebuild(j);
        else if (lane == 0) {
            WsBlk w = {~0ull, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
            blk[j] = w;
        }
    }
    __syncthreads();
    // Labels reach the instance map through a small LDS queue, flushed by all lanes every ~60 pops: gfx9 counts loads and stores in
    // ONE vmcnt, so a store per pop would make the next pop's key load wait for the store's round trip as well.
    int npend = 0;
    auto flush = [&]() {
        for (int e = lane; e < npend; e += 64) {
            const int q = (int)pend_i[e];
            const int ry = q / bw2;
            b.inst[g0 + (long)(y0 + ry - 1) * b.W + (x0 + q - ry * bw2 - 1)] = (int)pend_l[e];
        }
        npend = 0;
    };
    int live = top, nfree = 0;
    unsigned age = 0;
    bool tie = false;
    while (live > 0) {
        if (npend + 4 > NPEND) {
            flush();
            __syncthreads();
        }
        if (top + 4 > cap && nfree < 4) {
            if (lane == 0) atomicAdd(&b.stat[n].dbg[4], 1);
            tie = true;
            break;
        }
        const int nb = (top + 63) >> 6;
        u64 bk = ~0ull;
        unsigned ba = 0xffffffffu;
        int bj = 0;
        for (int j = lane; j < nb; j += 64) {
            const u64 k = blk[j].k;
            const unsigned a = blk[j].a;
            if (k < bk || (k == bk && a < ba)) {
                bk = k;
                ba = a;
                bj = j;
            }
        }
        const u64 gk = wave_min_u64(bk);
        const unsigned ga = wave_min_u32(bk == gk ? ba : 0xffffffffu);
        const u64 win = __ballot(bk == gk && ba == ga);
        if (ga == 0u) {                  // markers at the top: one of them alone, several of ONE label (harmless: ws_window_wave), or a real tie?
            unsigned cs = 0, l0 = 0xffffffffu, l1 = 0u;
            for (int j = lane; j < nb; j += 64)
                if (blk[j].k == gk && blk[j].a == 0u) {
                    cs += blk[j].c;
                    l0 = blk[j].l0 < l0 ? blk[j].l0 : l0;
                    l1 = blk[j].l1 > l1 ? blk[j].l1 : l1;
                }
            const u64 has = __ballot(cs > 0);
            if ((__popcll(has) > 1 || __any(cs > 1)) && wave_min_u32(l0) != ~wave_min_u32(~l1)) {
                if (lane == 0) atomicAdd(&b.stat[n].dbg[3], 1);
                tie = true;
                break;
            }
        }
        const int jb = __builtin_amdgcn_readlane(bj, __ffsll((long long)win) - 1);
        const int gs = (int)blk[jb].s;
        const int idx = (int)fi[gs];
        const unsigned lab = fl[gs];
        if (lane == 0) {                 // the slot dies and goes on the free stack
            fk[gs] = ~0ull;
            fa[gs] = 0xffffffffu;
            fs[nfree] = (unsigned)gs;
        }
        ++nfree;
        --live;
        __syncthreads();
        rebuild(jb);
        bool unl = false;
        int q = 0;
        if (lane < 4) {                  // skimage's neighbour order: up, left, right, down
            q = idx + (lane == 0 ? -bw2 : lane == 1 ? -1 : lane == 2 ? 1 : bw2);
            unl = bit(um, q) != 0;
        }
        const u64 umk = __ballot(unl);
        const int np = __popcll(umk);
        int slot = 0;
        u64 key = 0;
        unsigned pa = 0;
        if (unl) {
            const int r = __popcll(umk & ((1ull << lane) - 1ull));
            atomicAnd(&um[q >> 6], ~(1ull << (q & 63)));
            const int ry = q / bw2;
            pend_i[npend + r] = (unsigned)q;                                                  // labelled at push time
            pend_l[npend + r] = lab;
            slot = r < nfree ? (int)fs[nfree - 1 - r] : top + (r - nfree);
            key = kv[(ry - 1) * bw + (q - ry * bw2 - 1)];
            pa = age + 1u + (unsigned)r;
            fk[slot] = key;
            fa[slot] = pa;
            fi[slot] = (unsigned)q;
            fl[slot] = lab;
        }
        const int reuse = np < nfree ? np : nfree;
        nfree -= reuse;
        top += np - reuse;
        live += np;
        age += (unsigned)np;
        npend += np;
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 4; ++l)      // a pushed item may undercut its block's cached minimum (one at a time: two may share a block)
            if ((umk >> l) & 1ull) {
                const int sl = __builtin_amdgcn_readlane(slot, l);
                const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, l);
                const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), l);
                const u64 kk = ((u64)khi << 32) | klo;
                const unsigned aa = (unsigned)__builtin_amdgcn_readlane((int)pa, l);
                const int j = sl >> 6;
                const u64 ck = blk[j].k;
                const unsigned ca = blk[j].a;
                if ((kk < ck || (kk == ck && aa < ca)) && lane == 0) {
                    WsBlk w;
                    w.k = kk;
                    w.a = aa;
                    w.s = (unsigned)sl;
                    w.c = 1u;
                    w.l0 = w.l1 = lab;
                    w.pad = 0;
                    blk[j] = w;
                }
                __syncthreads();
            }
    }
    __syncthreads();
    flush();
    __syncthreads();
    return tie;
}

#define WS_AMAX 7600  // largest window replayed out of LDS (8-byte heap items: 20 B per pixel -> 152 KB)
#define WS_AMAX16 5400  // ... with the value inline in the heap item (28 B per pixel)
#define WS_LDS_BYTES (WS_AMAX * 20)
#define WS_MIN_CAP 1024  // oversized windows: at least this many heap slots (the top 10 levels) stay in LDS

// planes that are dead by now are re-used: par2 = ymin, hraw = ymax, vraw = xmin, cnt = xmax,
// lab = "component holds a marker", par = component list
__global__ __launch_bounds__(PP_T) void ws_init(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const long g = (long)n * b.P + i;
    b.inst[g] = 0;
    b.par2[g] = 0x7fffffff;
    ((int32_t *)b.hraw)[g] = -1;
    ((int32_t *)b.vraw)[g] = 0x7fffffff;
    b.cnt[g] = -1;
    b.lab[g] = 0;
}

__global__ __launch_bounds__(PP_T) void ws_bbox(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const long g0 = (long)n * b.P;
    const int r = b.broot[g0 + i];
    if (r < 0) return;
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    atomicMin(b.par2 + g0 + r, y);
    atomicMax((int32_t *)b.hraw + g0 + r, y);
    atomicMin((int32_t *)b.vraw + g0 + r, x);
    atomicMax(b.cnt + g0 + r, x);
    if (b.mk[g0 + i] > 0) b.lab[g0 + r] = 1;
}

__global__ __launch_bounds__(PP_T) void ws_list(PPBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= b.P) return;
    const long g0 = (long)n * b.P;
    if (b.broot[g0 + i] != (int)i || !b.lab[g0 + i]) return;
    const int slot = atomicAdd(&b.stat[n].n_comp, 1);
    b.par[g0 + slot] = (int)i;
    const int bh = ((const int32_t *)b.hraw)[g0 + i] - b.par2[g0 + i] + 1, bw = b.cnt[g0 + i] - ((const int32_t *)b.vraw)[g0 + i] + 1;
    atomicMax(&b.stat[n].max_area, bh * bw);
}

// Replay one window [y0..y0+bh) x [x0..x0+bw) of tile n; root >= 0 restricts the mask to that component,
// root < 0 takes the whole blob mask (whole-tile replay).  All 64 lanes stage / write back, lane 0 floods.
__device__ bool ws_window(PPBuf &b, int n, int root, int y0, int x0, int bh, int bw, unsigned char *lds, int *s_flag,
                          int lds_bytes = WS_LDS_BYTES)
{
    typedef unsigned long long u64;
    // a component whose bordered window fits the LDS is replayed by the whole wave; whole-tile replays (root < 0) ARE the
    // reference's tie order and keep the binary heap, as do windows beyond the LDS
    // (a marker tie has no defined order in the unsorted frontier: that component alone is redone below on the binary heap,
    // whose layout the reference's tie order follows, incl. the swap proof of a harmless two-way tie)
    long scratch_off = -1;       // scratch a failed wave attempt took: the binary-heap replay of the same component re-uses it
    if (root >= 0 && !b.no_wave) {
        const int A2 = (bh + 2) * (bw + 2);
        if ((long)28 * A2 <= lds_bytes) {
            if (threadIdx.x == 0) atomicAdd(&b.stat[n].dbg[0], 1);
            if (!ws_window_wave<false>(b, n, root, y0, x0, bh, bw, nullptr, nullptr, A2)) return false;
        } else if (lds_bytes >= WS_LDS_BYTES) {
            // window beyond the LDS (a clump of many nuclei, a tile-filling blob)
            const bool bitmap = (long)16 * ((A2 + 63) >> 6) + 20 * 2048 <= lds_bytes && !b.no_bitmap;
            const int need = bitmap ? bh * bw : A2;       // scratch entries (dead planes of this tile; a window is L2-resident)
            if (threadIdx.x == 0) *s_flag = atomicAdd(&b.stat[n].heap_top, need);
            __syncthreads();
            const long off = *s_flag;
            __syncthreads();
            if (off + need <= b.P) {
                const long g0s = (long)n * b.P;
                if (need >= bh * bw) scratch_off = off;
                if (bitmap) {
                    // bitmap window + labelled frontier in LDS, value keys of the interior in HBM scratch
                    if (threadIdx.x == 0) atomicAdd(&b.stat[n].dbg[1], 1);
                    if (!ws_window_wave_bitmap(b, n, root, y0, x0, bh, bw, (u64 *)(b.dist + g0s) + off, lds_bytes)) return false;
                } else {
                    // keys and labels in HBM scratch, the frontier keeps all of the LDS
                    if (threadIdx.x == 0) atomicAdd(&b.stat[n].dbg[2], 1);
                    if (!ws_window_wave<true>(b, n, root, y0, x0, bh, bw, (u64 *)(b.dist + g0s) + off, (int *)(b.overall + g0s) + off, lds_bytes / 16))
                        return false;
                }
            }
        }
    }
    if (threadIdx.x == 0) atomicAdd(&b.stat[n].dbg[root >= 0 ? 5 : 6], 1);
    const long g0 = (long)n * b.P;
    const int A = bh * bw;
    double *val;
    int32_t *out;
    u64 *heap;
    const bool in_lds = (long)20 * A <= lds_bytes;
    const bool inline_val = (long)28 * A <= lds_bytes;
    HItem *heap_far = nullptr;
    int cap = 0;
    if (in_lds) {
        val = (double *)lds;
        heap = (u64 *)(lds + (size_t)8 * A);
        out = (int32_t *)(lds + (size_t)(inline_val ? 24 : 16) * A);
    } else {
        // oversized window: values (and, beyond ~34k pixels, labels) stay in HBM scratch (dist / overall / heap planes are
        // dead by now); the top `cap` slots of the heap -- where every pop's sift-down runs -- and the label window live
        // in LDS, the deeper heap levels in HBM
        long off = scratch_off;
        if (off < 0) {
            if (threadIdx.x == 0) *s_flag = atomicAdd(&b.stat[n].heap_top, A);
            __syncthreads();
            off = *s_flag;
            __syncthreads();
        }
        if (off + A > b.P) return true;  // scratch exhausted (overlapping boxes): leave it to the whole-tile replay
        val = b.dist + g0 + off;
        heap_far = (HItem *)(b.heap + 2 * g0) + off;
        const size_t out_bytes = (size_t)4 * A;
        if (out_bytes + (size_t)16 * WS_MIN_CAP <= WS_LDS_BYTES) {
            out = (int32_t *)lds;
            heap = (u64 *)(lds + ((out_bytes + 15) & ~(size_t)15));
            cap = (int)((WS_LDS_BYTES - ((out_bytes + 15) & ~(size_t)15)) / 16);
        } else {
            out = (int32_t *)(b.overall + g0) + off;
            heap = (u64 *)lds;
            cap = WS_LDS_BYTES / 16;
        }
    }
    auto stage = [&]() {
        for (int t = threadIdx.x; t < A; t += 64) {
            const int yy = t / bw, xx = t - yy * bw;
            const long gi = g0 + (long)(y0 + yy) * b.W + (x0 + xx);
            const bool member = root >= 0 ? (b.broot[gi] == root) : (b.blb[gi] != 0);
            out[t] = member ? b.mk[gi] : -1;
            val[t] = b.blur[gi];
        }
        __syncthreads();
    };
    auto flood = [&](bool swap_first) {
        if (threadIdx.x == 0) {
            if (!in_lds) __threadfence();
            *s_flag = !in_lds ? ws_flood_window<2>(heap, val, out, bh, bw, swap_first, heap_far, cap)
                      : inline_val ? ws_flood_window<1>(heap, val, out, bh, bw, swap_first)
                                   : ws_flood_window<0>(heap, val, out, bh, bw, swap_first);
            if (!in_lds) __threadfence();
        }
        __syncthreads();
        const int r = *s_flag;
        __syncthreads();
        return r;
    };
    stage();
    const int ties = flood(false);
    for (int t = threadIdx.x; t < A; t += 64) {
        const int v = out[t];
        if (v >= 0) {
            const int yy = t / bw, xx = t - yy * bw;
            b.inst[g0 + (long)(y0 + yy) * b.W + (x0 + xx)] = v;
        }
    }
    if (root < 0 || ties == 0) return false;  // the whole-tile window IS the global run: its ties are skimage's
    if (ties >= 2) return true;
    // exactly one 2-way marker tie: replay with the two items exchanged; identical labels => harmless
    __syncthreads();
    stage();
    flood(true);
    int diff = 0;
    for (int t = threadIdx.x; t < A; t += 64) {
        const int v = out[t];
        if (v >= 0) {
            const int yy = t / bw, xx = t - yy * bw;
            diff |= b.inst[g0 + (long)(y0 + yy) * b.W + (x0 + xx)] != v;
        }
    }
    const bool any = __any(diff);
    __syncthreads();
    return any;
}

// Components are replayed in two launches by window size, because the LDS a workgroup declares sets how many of them a CU
// holds: `small` windows (<= WS_SMALL_A pixels: a single nucleus or a small clump, the bulk of a tile's components) take
// 28 KB, so five one-wave workgroups share a CU instead of one with the full 152 KB; the `large` launch (cls 1) takes the rest.
#define WS_SMALL_A 1024
#define WS_SMALL_LDS (WS_SMALL_A * 28)
__global__ __launch_bounds__(64) void ws_component(PPBuf b, int cls, int lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];
    __shared__ int s_flag;
    const int n = blockIdx.y;
    const long g0 = (long)n * b.P;
    const int ncomp = b.stat[n].n_comp;
    // Component k of tile n is worked by block (k + 5 n) mod gridDim.x.  Workgroups go to XCD (linear id mod 8) and gridDim.x is a
    // multiple of 8: without the rotation the first (often the only large) component of EVERY tile lands on the same XCD -- 64
    // tile-filling blobs then share 32 CUs and one 4 MB L2 (measured: 4.4 us per pop instead of 1 us).
    const int bx = (int)((blockIdx.x + gridDim.x - (5u * (unsigned)n) % gridDim.x) % gridDim.x);
    for (int k = bx; k < ncomp; k += gridDim.x) {
        const int root = b.par[g0 + k];
        const int y0 = b.par2[g0 + root], y1 = ((const int32_t *)b.hraw)[g0 + root];
        const int x0 = ((const int32_t *)b.vraw)[g0 + root], x1 = b.cnt[g0 + root];
        const int bh = y1 - y0 + 1, bw = x1 - x0 + 1;
        if (((long)(bh + 2) * (bw + 2) <= WS_SMALL_A ? 0 : 1) != cls) continue;   // uniform per workgroup
        if (ws_window(b, n, root, y0, x0, bh, bw, ws_lds, &s_flag, lds_bytes) && threadIdx.x == 0) b.stat[n].tie = 1;
    }
}

// tiles that reported a tie (or all tiles when forced): exact whole-tile replay = one window over the tile
__global__ __launch_bounds__(64) void ws_fallback(PPBuf b, int force)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];
    __shared__ int s_flag;
    const int n = blockIdx.x;
    if (!force && !b.stat[n].tie) return;
    if (threadIdx.x == 0) b.stat[n].heap_top = 0;  // the component pass is over: its HBM scratch is free again
    __syncthreads();
    ws_window(b, n, -1, 0, 0, b.H, b.W, ws_lds, &s_flag);
}

// ---------------------------------------------------------------------------------------------
static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t carve(PPBuf &b, unsigned char *base, int n, int H, int W)
{
    const size_t NP = (size_t)n * H * W;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char *p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    };
    b.stat = (TileStat *)take(sizeof(TileStat) * n);
    b.blb = (int32_t *)take(NP * 4);
    b.par = (int32_t *)take(NP * 4);
    b.cnt = (int32_t *)take(NP * 4);
    b.mk = (int32_t *)take(NP * 4);
    b.par2 = (int32_t *)take(NP * 4);
    b.lab = (int32_t *)take(NP * 4);
    b.broot = (int32_t *)take(NP * 4);
    b.bsz = (int32_t *)take(NP * 4);
    b.hraw = (float *)take(NP * 4);
    b.vraw = (float *)take(NP * 4);
    b.rowh = (double *)take(NP * 8);
    b.rowv = (double *)take(NP * 8);
    b.sobh = (double *)take(NP * 8);
    b.sobv = (double *)take(NP * 8);
    b.overall = (double *)take(NP * 8);
    b.dist = (double *)take(NP * 8);
    b.blur = (double *)take(NP * 8);
    b.m8a = (uint8_t *)take(NP);
    b.m8b = (uint8_t *)take(NP);
    b.heap = (unsigned long long *)take(NP * 16);
    return off;
}

__global__ void pp_stat_init(TileStat *s, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i].h_min = s[i].v_min = 0xffffffffu;
    s[i].h_max = s[i].v_max = 0u;
    s[i].sh_min = s[i].sv_min = ~0ull;
    s[i].sh_max = s[i].sv_max = 0ull;
    s[i].n_comp = 0;
    s[i].heap_top = 0;
    s[i].tie = 0;
    s[i].max_area = 0;
    for (int k = 0; k < 8; ++k) s[i].dbg[k] = 0;
}

static thread_local char pp_err[256] = "";
const char *hvn_pp_last_error() { return pp_err; }

static int postproc_impl(const float *pred, int n, int h, int w, int c, int c0, int32_t *inst, int32_t *tap_blb,
                         double *tap_dist, int32_t *tap_marker, void *workspace, size_t workspace_bytes, hipStream_t s)
{
    if (!pred || !inst || n <= 0 || h <= 0 || w <= 0 || c0 < 0 || c0 + 3 > c) return HVN_E_ARG;
    if ((long)h * w >= (1L << 31)) return HVN_E_ARG;
    PPBuf b;
    const size_t need = carve(b, (unsigned char *)workspace, n, h, w);
    if (!workspace || workspace_bytes < need) return HVN_E_SIZE;
    b.n = n; b.H = h; b.W = w; b.C = c; b.c0 = c0; b.P = (long)h * w;
    b.pred = pred; b.inst = inst;
    const dim3 grid((unsigned)((b.P + PP_T - 1) / PP_T), n), blk(PP_T);
    hipLaunchKernelGGL(pp_stat_init, dim3((n + 63) / 64), dim3(64), 0, s, b.stat, n);
    hipLaunchKernelGGL(pp_init, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_ccl_merge, grid, blk, 0, s, b.par, h, w, b.P);
    hipLaunchKernelGGL(pp_ccl_flatten_count, grid, blk, 0, s, b.par, b.cnt, b.P);
    hipLaunchKernelGGL(pp_blb_filter, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_sobel_row, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_sobel_col, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_combine, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_gauss_row, grid, blk, 0, s, b);
    hipMemsetAsync(b.cnt, 0, (size_t)n * b.P * 4, s);
    hipLaunchKernelGGL(pp_gauss_col_marker, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_ccl_merge, grid, blk, 0, s, b.par2, h, w, b.P);
    hipLaunchKernelGGL(pp_ccl_flatten_count, grid, blk, 0, s, b.par2, (int32_t *)nullptr, b.P);
    hipLaunchKernelGGL(pp_border_flag, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_fill, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_erode, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_dilate_init, grid, blk, 0, s, b);
    hipLaunchKernelGGL(pp_ccl_merge, grid, blk, 0, s, b.par, h, w, b.P);
    hipLaunchKernelGGL(pp_ccl_flatten_count, grid, blk, 0, s, b.par, b.cnt, b.P);
    hipLaunchKernelGGL(pp_rank_roots, dim3(n), dim3(1024), 0, s, b);
    hipLaunchKernelGGL(pp_marker_labels, grid, blk, 0, s, b);
    // taps first: the watershed stage recycles dead planes
    if (tap_marker) hipMemcpyAsync(tap_marker, b.mk, (size_t)n * b.P * 4, hipMemcpyDeviceToDevice, s);
    static int ws_mode = -1, ws_wave = 1, ws_bitmap = 1;  // HVN_WS_GLOBAL=1 forces the whole-tile replay everywhere (tests)
    if (ws_mode < 0) {
        const char *e = getenv("HVN_WS_GLOBAL");
        ws_mode = (e && atoi(e)) ? 1 : 0;
        e = getenv("HVN_WS_WAVE");
        ws_wave = e ? atoi(e) : 1;
        e = getenv("HVN_WS_BITMAP");
        ws_bitmap = e ? atoi(e) : 1;
    }
    b.no_wave = ws_wave ? 0 : 1;
    b.no_bitmap = ws_bitmap ? 0 : 1;
    hipLaunchKernelGGL(ws_init, grid, blk, 0, s, b);
    hipLaunchKernelGGL(ws_bbox, grid, blk, 0, s, b);
    hipLaunchKernelGGL(ws_list, grid, blk, 0, s, b);
    long maxc = b.P / 10 + 1;
    if (maxc > 2048) maxc = 2048;
    static std::atomic<unsigned long long> ws_attr_c{0}, ws_attr_f{0};     // per device (hvn_kernels.h)
    if (hvn_max_lds_once((const void *)ws_component, WS_LDS_BYTES, ws_attr_c) || hvn_max_lds_once((const void *)ws_fallback, WS_LDS_BYTES, ws_attr_f))
        return HVN_E_LAUNCH;
    if (!ws_mode) {
        hipLaunchKernelGGL(ws_component, dim3((unsigned)maxc, n), dim3(64), WS_SMALL_LDS, s, b, 0, WS_SMALL_LDS);
        hipLaunchKernelGGL(ws_component, dim3((unsigned)maxc, n), dim3(64), WS_LDS_BYTES, s, b, 1, WS_LDS_BYTES);
    }
    hipLaunchKernelGGL(ws_fallback, dim3(n), dim3(64), WS_LDS_BYTES, s, b, ws_mode);
    if (getenv("HVN_WS_STATS")) {   // diagnosis: which replay each component took (synchronous)
        hipStreamSynchronize(s);
        std::vector<TileStat> hs(n);
        hipMemcpy(hs.data(), b.stat, sizeof(TileStat) * (size_t)n, hipMemcpyDeviceToHost);
        long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0}, comps = 0, tiles_tie = 0;
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < 8; ++k) tot[k] += hs[i].dbg[k];
            comps += hs[i].n_comp;
            tiles_tie += hs[i].tie != 0;
        }
        fprintf(stderr, "[ws] %d tiles, %ld components: small-window %ld, bitmap-window %ld | handed over: mixed-label tie %ld, full frontier %ld | "
                        "component heap replays %ld, whole-tile replays %ld (tiles flagged %ld)\n",
                n, comps, tot[0], tot[1], tot[3], tot[4], tot[5], tot[6], tiles_tie);
    }
    const size_t NP = (size_t)n * b.P;
    if (tap_blb) hipMemcpyAsync(tap_blb, b.blb, NP * 4, hipMemcpyDeviceToDevice, s);
    if (tap_dist) hipMemcpyAsync(tap_dist, b.blur, NP * 8, hipMemcpyDeviceToDevice, s);
    return hipGetLastError() == hipSuccess ? HVN_OK : HVN_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// Per-instance table: the array half of post_proc.py:119-181 process() (bbox of
// misc/utils.py:18-28, cv2.moments m00/m10/m01 of the bbox crop, type majority vote).
struct InstAcc {
    int area, rmin, rmax, cmin, cmax, _pad;
    unsigned long long sx, sy;
};

__global__ __launch_bounds__(PP_T) void it_init(InstAcc *acc, int32_t *hist, long n_acc, long n_hist)
{
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i < n_acc) {
        InstAcc a;
        a.area = 0; a.rmin = 0x7fffffff; a.rmax = -1; a.cmin = 0x7fffffff; a.cmax = -1; a._pad = 0; a.sx = 0; a.sy = 0;
        acc[i] = a;
    }
    if (i < n_hist) hist[i] = 0;
}

__global__ __launch_bounds__(PP_T) void it_accumulate(const int32_t *inst, const float *pred, int H, int W, int C,
                                                      int nr_types, InstAcc *acc, int32_t *hist, int max_inst)
{
    const int n = blockIdx.y;
    const long P = (long)H * W;
    const long i = (long)blockIdx.x * PP_T + threadIdx.x;
    if (i >= P) return;
    const int l = inst[(long)n * P + i];
    if (l <= 0 || l > max_inst) return;
    const int y = (int)(i / W), x = (int)(i - (long)y * W);
    InstAcc *a = acc + (long)n * max_inst + (l - 1);
    atomicAdd(&a->area, 1);
    atomicMin(&a->rmin, y);
    atomicMax(&a->rmax, y);
    atomicMin(&a->cmin, x);
    atomicMax(&a->cmax, x);
    atomicAdd(&a->sx, (unsigned long long)x);
    atomicAdd(&a->sy, (unsigned long long)y);
    if (nr_types > 0) {
        int t = (int)pred[((long)n * P + i) * C];  // pred_type.astype(int32), post_proc.py:112
        if (t >= 0 && t < nr_types) atomicAdd(hist + ((long)n * max_inst + (l - 1)) * nr_types + t, 1);
    }
}

__global__ __launch_bounds__(PP_T) void it_finalize(const InstAcc *acc, const int32_t *hist, int nr_types, int max_inst,
                                                    hvn_inst_rec *rec, int32_t *counts)
{
    const int n = blockIdx.y;
    const int j = blockIdx.x * PP_T + threadIdx.x;
    if (j >= max_inst) return;
    const InstAcc a = acc[(long)n * max_inst + j];
    hvn_inst_rec r;
    r.label = j + 1;
    r.area = a.area;
    r.rmin = a.rmin; r.rmax = a.rmax + 1;  // get_bounding_box: rmax += 1, cmax += 1
    r.cmin = a.cmin; r.cmax = a.cmax + 1;
    r.sum_x = (double)(long long)(a.sx - (unsigned long long)a.area * (unsigned long long)(a.area ? a.cmin : 0));
    r.sum_y = (double)(long long)(a.sy - (unsigned long long)a.area * (unsigned long long)(a.area ? a.rmin : 0));
    r.type = -1;
    r.type_count = 0;
    if (a.area > 0 && nr_types > 0) {
        // sorted(by count, reverse=True) is stable over np.unique's ascending ids: ties -> smaller id
        const int32_t *h = hist + ((long)n * max_inst + j) * nr_types;
        int best = -1, bc = 0, best_nz = -1, bnz = 0, present = 0;
        for (int t = 0; t < nr_types; ++t) {
            const int c = h[t];
            if (c <= 0) continue;
            ++present;
            if (c > bc) { bc = c; best = t; }
            if (t != 0 && c > bnz) { bnz = c; best_nz = t; }
        }
        if (best == 0 && present > 1) { best = best_nz; bc = bnz; }  // post_proc.py:173-175
        r.type = best;
        r.type_count = bc;
    }
    rec[(long)n * max_inst + j] = r;
    if (a.area > 0) atomicAdd(counts + n, 1);
}

extern "C" {

size_t hvn_instance_table_workspace_bytes(int n, int max_inst, int nr_types)
{
    return align_up((size_t)n * max_inst * sizeof(InstAcc)) + align_up((size_t)n * max_inst * (nr_types > 0 ? nr_types : 1) * 4);
}

int hvn_instance_table(const int32_t *inst, const float *pred, int n, int h, int w, int c, int nr_types,
                       hvn_inst_rec *records, int32_t *counts, int max_inst, void *workspace, size_t workspace_bytes,
                       void *stream)
{
    if (!inst || !records || !counts || n <= 0 || h <= 0 || w <= 0 || max_inst <= 0 || nr_types < 0) return HVN_E_ARG;
    if (nr_types > 0 && !pred) return HVN_E_ARG;
    if (!workspace || workspace_bytes < hvn_instance_table_workspace_bytes(n, max_inst, nr_types)) return HVN_E_SIZE;
    hipStream_t s = (hipStream_t)stream;
    InstAcc *acc = (InstAcc *)workspace;
    int32_t *hist = (int32_t *)((unsigned char *)workspace + align_up((size_t)n * max_inst * sizeof(InstAcc)));
    const long n_acc = (long)n * max_inst, n_hist = n_acc * (nr_types > 0 ? nr_types : 1);
    const long m = n_acc > n_hist ? n_acc : n_hist;
    hipLaunchKernelGGL(it_init, dim3((unsigned)((m + PP_T - 1) / PP_T)), dim3(PP_T), 0, s, acc, hist, n_acc, n_hist);
    hipMemsetAsync(counts, 0, (size_t)n * 4, s);
    const long P = (long)h * w;
    hipLaunchKernelGGL(it_accumulate, dim3((unsigned)((P + PP_T - 1) / PP_T), n), dim3(PP_T), 0, s, inst, pred, h, w, c,
                       nr_types, acc, hist, max_inst);
    hipLaunchKernelGGL(it_finalize, dim3((max_inst + PP_T - 1) / PP_T, n), dim3(PP_T), 0, s, acc, hist, nr_types, max_inst,
                       records, counts);
    return hipGetLastError() == hipSuccess ? HVN_OK : HVN_E_LAUNCH;
}

size_t hvn_postproc_workspace_bytes(int n, int h, int w)
{
    PPBuf b;
    return carve(b, nullptr, n, h, w);
}

int hvn_postproc(const float *pred, int n, int h, int w, int c, int c0, int32_t *inst, void *workspace,
                 size_t workspace_bytes, void *stream)
{
    return postproc_impl(pred, n, h, w, c, c0, inst, nullptr, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

int hvn_postproc_taps(const float *pred, int n, int h, int w, int c, int c0, int32_t *inst, int32_t *blb, double *dist,
                      int32_t *marker, void *workspace, size_t workspace_bytes, void *stream)
{
    return postproc_impl(pred, n, h, w, c, c0, inst, blb, dist, marker, workspace, workspace_bytes, (hipStream_t)stream);
}

int hvn_postproc_stats(const void *workspace, size_t workspace_bytes, int n, int h, int w, long long out[10], void *stream)
{
    if (!workspace || !out || n <= 0 || h <= 0 || w <= 0) return HVN_E_ARG;
    if (workspace_bytes < hvn_postproc_workspace_bytes(n, h, w)) return HVN_E_SIZE;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return HVN_E_LAUNCH;
    std::vector<TileStat> hs(n);            // the per-map reduction slots lead the workspace (carve)
    if (hipMemcpy(hs.data(), workspace, sizeof(TileStat) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return HVN_E_LAUNCH;
    for (int k = 0; k < 10; ++k) out[k] = 0;
    for (int i = 0; i < n; ++i) {
        out[0] += hs[i].n_comp;
        out[1] += hs[i].dbg[0]; out[2] += hs[i].dbg[1]; out[3] += hs[i].dbg[2];
        out[4] += hs[i].dbg[3]; out[5] += hs[i].dbg[4]; out[6] += hs[i].dbg[5]; out[7] += hs[i].dbg[6];
        out[8] += hs[i].tie != 0;
        if (hs[i].max_area > out[9]) out[9] = hs[i].max_area;
    }
    return HVN_OK;
}

}  // extern "C"
