/*
 * hvn_oracle.c -- CPU restatement of the HoVer-Net instance-separation step.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and only as the checker / the
 * timed CPU baseline.  The product path (hover_net_amd/) never links, imports
 * or calls anything in oracle/.
 *
 * What it restates (reference = vqdang/hover_net, /root/reference):
 *   models/hovernet/post_proc.py:26-90   __proc_np_hv      (P1..P11 of SURVEY 8a)
 *   misc/utils.py:142-182                remove_small_objects
 * and the third-party arithmetic those lines call into, none of which lives
 * under /root/reference:
 *   opencv-python==4.3.0.36 (requirements.txt:6)  normalize / Sobel /
 *       GaussianBlur / getStructuringElement / morphologyEx
 *   scipy==1.5.2 (requirements.txt:12)            ndimage.label, binary_fill_holes
 *   scikit-image==0.17.2 (requirements.txt:10)    segmentation.watershed
 *
 * Pinning status:
 *   - scipy / skimage parts + numpy dtype glue: PINNED.  tests/golden/pp_*.npz are
 *     produced by the reference's own, unmodified post_proc.py running under
 *     /opt/conda/bin/python3.9 with real scipy 1.7.1 + scikit-image 0.18.3
 *     (oracle/make_golden_postproc.py); this file must reproduce them bit for bit.
 *   - OpenCV parts: pinned to a SECOND, independently written restatement, not to OpenCV itself.
 *     OpenCV is not installed anywhere on the build box, so the cv2 arithmetic below follows the
 *     OpenCV 4.3 sources from memory (file names cited at each function), and the golden files were
 *     made with oracle/cv2_shim/cv2.py, which calls *this* library for normalize / Sobel /
 *     GaussianBlur / morphologyEx.  What guards against a mis-remembered kernel, border mode or
 *     structuring element: oracle/cv2_shim_scipy/cv2.py restates the same calls with scipy.ndimage in
 *     float64 without this file's operation order; tests/test_oracle_cv2_independent.py holds the two
 *     together (integer results bit-equal, floating point within a few ulp = summation order only) and
 *     oracle/check_alt_shim.py shows that the reference's own __proc_np_hv over the second shim
 *     reproduces all 56 committed golden instance maps exactly (reference PQ == 1).  What stays
 *     unverifiable without OpenCV: the last-bit rounding ORDER of its SIMD paths.
 *     Assumed build: the AVX2/FMA3 dispatch units (what an opencv-python wheel
 *     runs on any x86 host since Haswell), i.e. `s += f*x` contracts to one fma.
 *     Where the products are exact (integer taps x float32 data, power-of-two
 *     taps) this makes no difference; it matters in the Sobel column pass and
 *     in convertTo.
 *   - process()'s per-instance loop (post_proc.py:119-181; cv2.moments / findContours): restated in
 *     python (oracle/process_np.py + oracle/cv2_shim/_suzuki.py), pinned by tests/golden/proc_*.npz,
 *     which the reference's own process() made (oracle/make_golden_process.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared  (see oracle/Makefile).
 * -ffp-contract=off so that the only fused operations are the explicit fma()/
 * fmaf() calls below.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HVN_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* border handling: cv::borderInterpolate(p, len, BORDER_REFLECT_101)         */
/* (OpenCV core/src/copy.cpp)                                                 */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else       p = 2 * len - 2 - p;
    }
    return p;
}

/* ------------------------------------------------------------------------- */
/* P4: cv2.normalize(src32f, None, 0, 1, NORM_MINMAX, dtype=CV_32F)           */
/* post_proc.py:49-54.  OpenCV core/src/norm.cpp normalize():                 */
/*   scale = (dmax-dmin) * (smax-smin > DBL_EPSILON ? 1./(smax-smin) : 0)     */
/*   rtype==CV_32F: scale=(float)scale; shift=(float)dmin-(float)(smin*scale) */
/* then Mat::convertTo -> convert_scale.simd.hpp cvt_32f: v_fma(src, a, b).   */
HVN_API void hvn_o_minmax_32f(const float *src, size_t n, double *mn, double *mx)
{
    float lo = src[0], hi = src[0];
    for (size_t i = 1; i < n; ++i) {
        if (src[i] < lo) lo = src[i];
        if (src[i] > hi) hi = src[i];
    }
    *mn = (double)lo;
    *mx = (double)hi;
}

HVN_API void hvn_o_norm_coeffs(double smin, double smax, float *a, float *b)
{
    double scale = 1.0 * ((smax - smin > DBL_EPSILON) ? 1. / (smax - smin) : 0.);
    scale = (double)(float)scale;
    double shift = (double)((float)0.0 - (float)(smin * scale));
    *a = (float)scale;
    *b = (float)shift;
}

HVN_API void hvn_o_normalize_32f(const float *src, float *dst, size_t n)
{
    double mn, mx;
    float a, b;
    hvn_o_minmax_32f(src, n, &mn, &mx);
    hvn_o_norm_coeffs(mn, mx, &a, &b);
    for (size_t i = 0; i < n; ++i) dst[i] = fmaf(src[i], a, b);
}

/* P6: cv2.normalize(src64f, None, 0, 1, NORM_MINMAX, dtype=CV_32F)           */
/* post_proc.py:59-68.  Same coefficients; convertTo 64f->32f is cvt_64f:     */
/* v_fma in double, then narrowing store.                                     */
HVN_API void hvn_o_minmax_64f(const double *src, size_t n, double *mn, double *mx)
{
    double lo = src[0], hi = src[0];
    for (size_t i = 1; i < n; ++i) {
        if (src[i] < lo) lo = src[i];
        if (src[i] > hi) hi = src[i];
    }
    *mn = lo;
    *mx = hi;
}

HVN_API void hvn_o_normalize_64f32f(const double *src, float *dst, size_t n)
{
    double mn, mx;
    float a, b;
    hvn_o_minmax_64f(src, n, &mn, &mx);
    hvn_o_norm_coeffs(mn, mx, &a, &b);
    for (size_t i = 0; i < n; ++i) dst[i] = (float)fma(src[i], (double)a, (double)b);
}

/* ------------------------------------------------------------------------- */
/* P5: cv2.Sobel(src32f, CV_64F, dx, dy, ksize=21), post_proc.py:56-57        */
/* OpenCV imgproc/src/deriv.cpp getSobelKernels: the order-0 kernel is the    */
/* binomial row C(20,k); the order-1 kernel is C(19,k-1)-C(19,k).  Generated  */
/* by the same repeated [1 1] / [-1 1] convolution OpenCV uses.               */
static void sobel_kernel21(int order, double *k /*21*/)
{
    int ksize = 21;
    int kerI[22];
    int i, j;
    kerI[0] = 1;
    for (i = 0; i < ksize; i++) kerI[i + 1] = 0;
    for (i = 0; i < ksize - order - 1; i++) {
        int oldval = kerI[0];
        for (j = 1; j <= ksize; j++) {
            int newval = kerI[j] + kerI[j - 1];
            kerI[j - 1] = oldval;
            oldval = newval;
        }
    }
    for (i = 0; i < order; i++) {
        int oldval = -kerI[0];
        for (j = 1; j <= ksize; j++) {
            int newval = kerI[j - 1] - kerI[j];
            kerI[j - 1] = oldval;
            oldval = newval;
        }
    }
    for (i = 0; i < ksize; i++) k[i] = (double)kerI[i];
}

HVN_API void hvn_o_sobel_kernel21(int order, double *k) { sobel_kernel21(order, k); }

/* dx=1: derivative along x (row pass uses the derivative kernel, column pass  */
/* the smoothing kernel); dx=0: derivative along y.                            */
/* Row pass   = filter.simd.hpp RowFilter<float,double,RowNoVec>:              */
/*              s = kx[0]*S[0]; for k=1..20: s += kx[k]*S[k]   (products exact) */
/* Column pass= SymmColumnFilter<Cast<double,double>,ColumnNoVec>:             */
/*   symmetric : s = ky[0]*C;  for k=1..10: s = fma(ky[k], (R[+k] + R[-k]), s)  */
/*   asymmetric: s = 0;        for k=1..10: s = fma(ky[k], (R[+k] - R[-k]), s)  */
/*   (ky indexed from the kernel centre, +k = rows below).                     */
HVN_API void hvn_o_sobel21(const float *src, double *dst, int H, int W, int dx)
{
    double kd[21], ks[21];
    sobel_kernel21(1, kd);
    sobel_kernel21(0, ks);
    const double *kx = dx ? kd : ks;
    const double *ky = dx ? ks : kd;
    const int R = 10;
    double *row = (double *)malloc(sizeof(double) * (size_t)H * W);
    for (int y = 0; y < H; ++y) {
        const float *S = src + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            double s = kx[0] * (double)S[reflect101(x - R, W)];
            for (int k = 1; k < 21; ++k) s += kx[k] * (double)S[reflect101(x - R + k, W)];
            row[(size_t)y * W + x] = s;
        }
    }
    const int symm = dx ? 1 : 0; /* column kernel is the smoothing one when dx=1 */
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            double s;
            if (symm) {
                s = ky[R] * row[(size_t)y * W + x] + 0.0;
                for (int k = 1; k <= R; ++k) {
                    double a = row[(size_t)reflect101(y + k, H) * W + x];
                    double b = row[(size_t)reflect101(y - k, H) * W + x];
                    s = fma(ky[R + k], a + b, s);
                }
            } else {
                s = 0.0;
                for (int k = 1; k <= R; ++k) {
                    double a = row[(size_t)reflect101(y + k, H) * W + x];
                    double b = row[(size_t)reflect101(y - k, H) * W + x];
                    s = fma(ky[R + k], a - b, s);
                }
            }
            dst[(size_t)y * W + x] = s;
        }
    }
    free(row);
}

/* ------------------------------------------------------------------------- */
/* P8: cv2.GaussianBlur(src64f, (3,3), 0), post_proc.py:76                    */
/* smooth.dispatch.cpp getGaussianKernel: ksize 3, sigma<=0 -> fixed taps      */
/* {0.25, 0.5, 0.25}; sepFilter2D in CV_64F, BORDER_REFLECT_101.               */
/* Row: RowFilter<double,double>: (0.25*l + 0.5*c) + 0.25*r                    */
/* Col: SymmColumnFilter: 0.5*c + 0.25*(below + above)                         */
HVN_API void hvn_o_gauss3_64f(const double *src, double *dst, int H, int W)
{
    double *row = (double *)malloc(sizeof(double) * (size_t)H * W);
    for (int y = 0; y < H; ++y) {
        const double *S = src + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            double s = 0.25 * S[reflect101(x - 1, W)];
            s += 0.5 * S[x];
            s += 0.25 * S[reflect101(x + 1, W)];
            row[(size_t)y * W + x] = s;
        }
    }
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            double s = 0.5 * row[(size_t)y * W + x] + 0.0;
            double a = row[(size_t)reflect101(y + 1, H) * W + x];
            double b = row[(size_t)reflect101(y - 1, H) * W + x];
            s = fma(0.25, a + b, s);
            dst[(size_t)y * W + x] = s;
        }
    }
    free(row);
}

/* ------------------------------------------------------------------------- */
/* P10: cv2.morphologyEx(m8u, MORPH_OPEN, getStructuringElement(ELLIPSE,5x5)) */
/* post_proc.py:83-84.  morph.dispatch.cpp: element rows 00100/11111x3/00100, */
/* erode then dilate, BORDER_CONSTANT with morphologyDefaultBorderValue():    */
/* out-of-image pixels never win the min (erode) nor the max (dilate).        */
static const int8_t ELL5[5][5] = {
    {0, 0, 1, 0, 0}, {1, 1, 1, 1, 1}, {1, 1, 1, 1, 1}, {1, 1, 1, 1, 1}, {0, 0, 1, 0, 0}};

HVN_API void hvn_o_morph_open5(const uint8_t *src, uint8_t *dst, int H, int W)
{
    uint8_t *er = (uint8_t *)malloc((size_t)H * W);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t m = 255;
            for (int j = -2; j <= 2; ++j)
                for (int i = -2; i <= 2; ++i) {
                    if (!ELL5[j + 2][i + 2]) continue;
                    int yy = y + j, xx = x + i;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    uint8_t v = src[(size_t)yy * W + xx];
                    if (v < m) m = v;
                }
            er[(size_t)y * W + x] = m;
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t m = 0;
            for (int j = -2; j <= 2; ++j)
                for (int i = -2; i <= 2; ++i) {
                    if (!ELL5[j + 2][i + 2]) continue;
                    int yy = y + j, xx = x + i;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    uint8_t v = er[(size_t)yy * W + xx];
                    if (v > m) m = v;
                }
            dst[(size_t)y * W + x] = m;
        }
    free(er);
}

/* ------------------------------------------------------------------------- */
/* P2: scipy.ndimage.label(bin) with the default (4-connected) structure,      */
/* post_proc.py:45,85.  Labels are numbered in raster order of each            */
/* component's first pixel (checked against scipy by the golden files).        */
static int32_t uf_find(int32_t *p, int32_t i)
{
    while (p[i] != i) {
        p[i] = p[p[i]];
        i = p[i];
    }
    return i;
}

HVN_API int hvn_o_label4(const int32_t *bin, int32_t *lab, int H, int W)
{
    size_t n = (size_t)H * W;
    int32_t *par = (int32_t *)malloc(sizeof(int32_t) * n);
    for (size_t i = 0; i < n; ++i) par[i] = (int32_t)i;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)y * W + x;
            if (!bin[i]) continue;
            if (x > 0 && bin[i - 1]) {
                int32_t a = uf_find(par, (int32_t)i), b = uf_find(par, (int32_t)(i - 1));
                if (a < b) par[b] = a; else par[a] = b;
            }
            if (y > 0 && bin[i - W]) {
                int32_t a = uf_find(par, (int32_t)i), b = uf_find(par, (int32_t)(i - W));
                if (a < b) par[b] = a; else par[a] = b;
            }
        }
    int32_t next = 0;
    /* the root of every component is its smallest raster index, so visiting   */
    /* pixels in raster order meets roots in first-pixel order                 */
    for (size_t i = 0; i < n; ++i) {
        if (!bin[i]) { lab[i] = 0; continue; }
        int32_t r = uf_find(par, (int32_t)i);
        if ((size_t)r == i) lab[i] = ++next;
        else lab[i] = lab[r];
    }
    free(par);
    return next;
}

/* P3: misc/utils.py:142-182 remove_small_objects on an int label image.       */
HVN_API void hvn_o_remove_small(int32_t *lab, size_t n, int nlabels, int min_size)
{
    int64_t *cnt = (int64_t *)calloc((size_t)nlabels + 1, sizeof(int64_t));
    for (size_t i = 0; i < n; ++i) cnt[lab[i]]++;
    for (size_t i = 0; i < n; ++i)
        if (cnt[lab[i]] < min_size) lab[i] = 0;
    free(cnt);
}

/* P9: scipy.ndimage.binary_fill_holes(m) (default 4-connected structure):     */
/* a background pixel stays background iff it is 4-connected to the border.    */
HVN_API void hvn_o_fill_holes(const int32_t *m, uint8_t *out, int H, int W)
{
    size_t n = (size_t)H * W;
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * n);
    size_t sp = 0;
    for (size_t i = 0; i < n; ++i) out[i] = 1;
#define PUSH_BG(ii)                                   \
    do {                                              \
        size_t _i = (ii);                             \
        if (!m[_i] && out[_i]) { out[_i] = 0; stack[sp++] = (int32_t)_i; } \
    } while (0)
    for (int x = 0; x < W; ++x) { PUSH_BG((size_t)x); PUSH_BG((size_t)(H - 1) * W + x); }
    for (int y = 0; y < H; ++y) { PUSH_BG((size_t)y * W); PUSH_BG((size_t)y * W + W - 1); }
    while (sp) {
        int32_t i = stack[--sp];
        int y = i / W, x = i % W;
        if (x > 0) PUSH_BG((size_t)i - 1);
        if (x < W - 1) PUSH_BG((size_t)i + 1);
        if (y > 0) PUSH_BG((size_t)i - W);
        if (y < H - 1) PUSH_BG((size_t)i + W);
    }
#undef PUSH_BG
    free(stack);
}

/* ------------------------------------------------------------------------- */
/* P11: skimage.segmentation.watershed(image, markers, mask=mask),             */
/* post_proc.py:88; connectivity 1, compactness 0, no watershed line.          */
/* Model = SURVEY.md Appendix B (skimage/segmentation/_watershed.py:204-231    */
/* + _watershed_cy.pyx + _shared/heap_general.pxi of 0.17/0.18).               */
typedef struct { double value; int32_t age; int32_t index; } hitem;

static inline int smaller(const hitem *a, const hitem *b)
{
    if (a->value != b->value) return a->value < b->value;
    return a->age < b->age;
}

typedef struct { hitem *d; size_t n, cap; } heap_t;

static void heap_push(heap_t *h, hitem it)
{
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 1024;
        h->d = (hitem *)realloc(h->d, h->cap * sizeof(hitem));
    }
    size_t child = h->n++;
    h->d[child] = it;
    while (child > 0) {
        size_t parent = (child + 1) / 2 - 1;
        if (smaller(&h->d[child], &h->d[parent])) {
            hitem t = h->d[child]; h->d[child] = h->d[parent]; h->d[parent] = t;
            child = parent;
        } else break;
    }
}

static hitem heap_pop(heap_t *h)
{
    hitem top = h->d[0];
    h->d[0] = h->d[--h->n];
    size_t i = 0, n = h->n;
    for (;;) {
        size_t l = 2 * i + 1, r = 2 * i + 2, s = i;
        if (l < n) {
            if (smaller(&h->d[l], &h->d[i])) s = l;
            if (r < n && smaller(&h->d[r], &h->d[s])) s = r;
        } else break;
        if (s == i) break;
        hitem t = h->d[i]; h->d[i] = h->d[s]; h->d[s] = t;
        i = s;
    }
    return top;
}

HVN_API void hvn_o_watershed(const double *image, const int32_t *markers, const int32_t *mask,
                             int32_t *out, int H, int W)
{
    const int Wp = W + 2, Hp = H + 2;
    size_t np = (size_t)Wp * Hp;
    double *img = (double *)calloc(np, sizeof(double));
    int8_t *msk = (int8_t *)calloc(np, 1);
    int32_t *o = (int32_t *)calloc(np, sizeof(int32_t));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t s = (size_t)y * W + x, d = (size_t)(y + 1) * Wp + (x + 1);
            img[d] = image[s];
            msk[d] = mask[s] != 0;
            o[d] = msk[d] ? markers[s] : 0; /* markers * mask (_watershed.py:84) */
        }
    const int nb[4] = {-Wp, -1, 1, Wp};
    heap_t h = {0, 0, 0};
    for (size_t i = 0; i < np; ++i)
        if (o[i]) { hitem it = {img[i], 0, (int32_t)i}; heap_push(&h, it); }
    int32_t age = 0;
    while (h.n) {
        hitem e = heap_pop(&h);
        for (int k = 0; k < 4; ++k) {
            int32_t n = e.index + nb[k];
            if (!msk[n] || o[n]) continue;
            age += 1;
            o[n] = o[e.index];
            hitem it = {img[n], age, n};
            heap_push(&h, it);
        }
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) out[(size_t)y * W + x] = o[(size_t)(y + 1) * Wp + (x + 1)];
    free(h.d); free(img); free(msk); free(o);
}

/* ------------------------------------------------------------------------- */
/* Whole chain: post_proc.py:26-90.  pred is H x W x 3 float32 [p, h, v].       */
/* Optional taps (may be NULL) expose intermediates for stage-wise GPU tests.  */
HVN_API void hvn_o_proc_np_hv_ex(const float *pred, int H, int W, int32_t *out,
                                 int32_t *tap_blb, double *tap_dist, int32_t *tap_marker)
{
    size_t n = (size_t)H * W;
    float *hraw = (float *)malloc(n * 4), *vraw = (float *)malloc(n * 4);
    float *hn = (float *)malloc(n * 4), *vn = (float *)malloc(n * 4);
    float *sh = (float *)malloc(n * 4), *sv = (float *)malloc(n * 4);
    double *sob = (double *)malloc(n * 8), *dist = (double *)malloc(n * 8);
    double *overall = (double *)malloc(n * 8), *blur = (double *)malloc(n * 8);
    int32_t *blb = (int32_t *)malloc(n * 4), *lab = (int32_t *)malloc(n * 4);
    int32_t *mk = (int32_t *)malloc(n * 4);
    uint8_t *m8 = (uint8_t *)malloc(n), *m8o = (uint8_t *)malloc(n);

    for (size_t i = 0; i < n; ++i) {
        blb[i] = pred[3 * i] >= 0.5f;            /* :43 */
        hraw[i] = pred[3 * i + 1];
        vraw[i] = pred[3 * i + 2];
    }
    int nl = hvn_o_label4(blb, lab, H, W);       /* :45 */
    hvn_o_remove_small(lab, n, nl, 10);          /* :46 */
    for (size_t i = 0; i < n; ++i) blb[i] = lab[i] > 0; /* :47 */

    hvn_o_normalize_32f(hraw, hn, n);            /* :49-54 */
    hvn_o_normalize_32f(vraw, vn, n);
    hvn_o_sobel21(hn, sob, H, W, 1);             /* :56 */
    hvn_o_normalize_64f32f(sob, sh, n);          /* :59-63 */
    hvn_o_sobel21(vn, sob, H, W, 0);             /* :57 */
    hvn_o_normalize_64f32f(sob, sv, n);          /* :64-68 */
    for (size_t i = 0; i < n; ++i) {
        float a = 1.0f - sh[i], b = 1.0f - sv[i];     /* float32 arithmetic */
        float m = a > b ? a : b;                      /* :70 np.maximum */
        double ov = (double)m - (double)(1 - blb[i]); /* :71 float32 - int32 -> float64 */
        if (ov < 0) ov = 0;                           /* :72 */
        overall[i] = ov;
        dist[i] = (1.0 - ov) * (double)blb[i];        /* :74 */
    }
    hvn_o_gauss3_64f(dist, blur, H, W);          /* :76 */
    for (size_t i = 0; i < n; ++i) {
        blur[i] = -blur[i];
        int32_t ovb = overall[i] >= 0.4;         /* :78 */
        int32_t m = blb[i] - ovb;                /* :80 */
        mk[i] = m < 0 ? 0 : m;                   /* :81 */
    }
    hvn_o_fill_holes(mk, m8, H, W);              /* :82 */
    hvn_o_morph_open5(m8, m8o, H, W);            /* :83-84 */
    for (size_t i = 0; i < n; ++i) mk[i] = m8o[i];
    nl = hvn_o_label4(mk, lab, H, W);            /* :85 */
    hvn_o_remove_small(lab, n, nl, 10);          /* :86 */
    hvn_o_watershed(blur, lab, blb, out, H, W);  /* :88 */

    if (tap_blb) memcpy(tap_blb, blb, n * 4);
    if (tap_dist) memcpy(tap_dist, blur, n * 8);
    if (tap_marker) memcpy(tap_marker, lab, n * 4);
    free(hraw); free(vraw); free(hn); free(vn); free(sh); free(sv); free(sob); free(dist);
    free(overall); free(blur); free(blb); free(lab); free(mk); free(m8); free(m8o);
}

HVN_API void hvn_o_proc_np_hv(const float *pred, int H, int W, int32_t *out)
{
    hvn_o_proc_np_hv_ex(pred, H, W, out, NULL, NULL, NULL);
}

/* batch helper for the cpu_baseline leg of bench.py: n maps of H x W x C,     */
/* channel offset c0 (1 when a type channel leads).                            */
HVN_API void hvn_o_proc_batch(const float *pred, int n, int H, int W, int C, int c0, int32_t *out)
{
    size_t px = (size_t)H * W;
    float *tmp = (float *)malloc(px * 3 * sizeof(float));
    for (int b = 0; b < n; ++b) {
        const float *p = pred + (size_t)b * px * C;
        for (size_t i = 0; i < px; ++i) {
            tmp[3 * i] = p[i * C + c0];
            tmp[3 * i + 1] = p[i * C + c0 + 1];
            tmp[3 * i + 2] = p[i * C + c0 + 2];
        }
        hvn_o_proc_np_hv(tmp, H, W, out + (size_t)b * px);
    }
    free(tmp);
}
