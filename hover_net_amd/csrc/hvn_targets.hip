// hvn_targets.hip -- training-target generation on the GPU:
// /root/reference/models/hovernet/targets.py:17-116 gen_instance_hv_map / gen_targets with its helpers
// fix_mirror_padding (dataloader/augs.py:18-32), cropping_center (misc/utils.py:32-52) and
// skimage.morphology.remove_small_objects(min_size=30) on the cropped label map.
//
// Per image of a batch: (1) instances = 4-connected components of equal annotation id (lock-free union-find, root =
// smallest raster index) -- what fix_mirror_padding's relabelling amounts to; (2) per instance: pixel count inside the
// centre crop, bounding box, coordinate sums (centre of mass), by atomics; (3) an instance gets HV targets iff it has
// >= 30 pixels inside the crop and its box widened by 2 px does not start before the image (the reference slices with
// a negative start there, which yields an empty crop: the instance is skipped); (4) offsets from the rounded centre of
// mass, extrema of the negative / positive side per instance (atomicMin / Max on ints), normalised in float32 exactly as
// numpy does (IEEE single division); (5) the centre crop of the HV map and of (ann > 0) is written.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvn_kernels.h"

#define TG_T 256

struct TgBuf {
    const int32_t *ann;
    int32_t *par, *area, *carea, *rmin, *rmax, *cmin, *cmax, *comr, *comc, *minx, *maxx, *miny, *maxy;
    unsigned long long *sumr, *sumc;
    float *hv;
    int32_t *np_map;
    int N, H, W, CH, CW, h0, w0;
    long P;
};

__device__ inline int tg_find(int32_t *par, int i)
{
    int r = i;
    while (true) {
        const int p = par[r];
        if (p == r) return r;
        r = p;
    }
}

__device__ inline void tg_union(int32_t *par, int a, int b)
{
    while (true) {
        a = tg_find(par, a);
        b = tg_find(par, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }  // a > b: hang a under b
        const int old = atomicMin(par + a, b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(TG_T) void tg_init(TgBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * TG_T + threadIdx.x;
    if (i >= b.P) return;
    const long g = (long)n * b.P + i;
    b.par[g] = b.ann[g] > 0 ? (int)i : -1;
    b.area[g] = 0;
    b.carea[g] = 0;
    b.rmin[g] = 0x7fffffff;
    b.cmin[g] = 0x7fffffff;
    b.rmax[g] = -1;
    b.cmax[g] = -1;
    b.sumr[g] = 0;
    b.sumc[g] = 0;
    b.minx[g] = 0;
    b.maxx[g] = 0;
    b.miny[g] = 0;
    b.maxy[g] = 0;
}

__global__ __launch_bounds__(TG_T) void tg_merge(TgBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * TG_T + threadIdx.x;
    if (i >= b.P) return;
    const long g0 = (long)n * b.P;
    const int id = b.ann[g0 + i];
    if (id <= 0) return;
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    if (x + 1 < b.W && b.ann[g0 + i + 1] == id) tg_union(b.par + g0, (int)i, (int)i + 1);
    if (y + 1 < b.H && b.ann[g0 + i + b.W] == id) tg_union(b.par + g0, (int)i, (int)i + b.W);
}

__global__ __launch_bounds__(TG_T) void tg_stats(TgBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * TG_T + threadIdx.x;
    if (i >= b.P) return;
    const long g0 = (long)n * b.P;
    if (b.par[g0 + i] < 0) return;
    const int r = tg_find(b.par + g0, (int)i);
    b.par[g0 + i] = r;   // flatten (roots are fixed points, so concurrent finds stay correct)
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    atomicAdd(b.area + g0 + r, 1);
    if (y >= b.h0 && y < b.h0 + b.CH && x >= b.w0 && x < b.w0 + b.CW) atomicAdd(b.carea + g0 + r, 1);
    atomicMin(b.rmin + g0 + r, y);
    atomicMax(b.rmax + g0 + r, y);
    atomicMin(b.cmin + g0 + r, x);
    atomicMax(b.cmax + g0 + r, x);
    atomicAdd(b.sumr + g0 + r, (unsigned long long)y);
    atomicAdd(b.sumc + g0 + r, (unsigned long long)x);
}

// per root: eligibility and the rounded centre of mass in widened-box coordinates (-1: no HV target)
__global__ __launch_bounds__(TG_T) void tg_roots(TgBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * TG_T + threadIdx.x;
    if (i >= b.P) return;
    const long g = (long)n * b.P + i;
    if (b.par[g] != (int)i) return;
    const int r0 = b.rmin[g] - 2, c0 = b.cmin[g] - 2;
    if (b.carea[g] < 30 || r0 < 0 || c0 < 0) {
        b.comr[g] = -1;
        return;
    }
    const long long area = b.area[g];
    // scipy center_of_mass over the widened box: sum(local coord * m) / sum(m) in float64 (integer numerator: exact)
    b.comr[g] = (int)((double)((long long)b.sumr[g] - (long long)r0 * area) / (double)area + 0.5);
    b.comc[g] = (int)((double)((long long)b.sumc[g] - (long long)c0 * area) / (double)area + 0.5);
}

__device__ inline bool tg_offsets(const TgBuf &b, long g0, long i, int &r, int &xo, int &yo)
{
    r = b.par[g0 + i];
    if (r < 0 || b.comr[g0 + r] < 0) return false;
    const int y = (int)(i / b.W), x = (int)(i - (long)y * b.W);
    xo = (x - (b.cmin[g0 + r] - 2)) + 1 - b.comc[g0 + r];   // np.arange(1, w + 1) - com
    yo = (y - (b.rmin[g0 + r] - 2)) + 1 - b.comr[g0 + r];
    return true;
}

__global__ __launch_bounds__(TG_T) void tg_extrema(TgBuf b)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * TG_T + threadIdx.x;
    if (i >= b.P) return;
    const long g0 = (long)n * b.P;
    int r, xo, yo;
    if (!tg_offsets(b, g0, i, r, xo, yo)) return;
    if (xo < 0) atomicMin(b.minx + g0 + r, xo);
    if (xo > 0) atomicMax(b.maxx + g0 + r, xo);
    if (yo < 0) atomicMin(b.miny + g0 + r, yo);
    if (yo > 0) atomicMax(b.maxy + g0 + r, yo);
}

__global__ __launch_bounds__(TG_T) void tg_write(TgBuf b)
{
    const int n = blockIdx.y;
    const long j = (long)blockIdx.x * TG_T + threadIdx.x;   // index inside the crop
    if (j >= (long)b.CH * b.CW) return;
    const int cy = (int)(j / b.CW), cx = (int)(j - (long)cy * b.CW);
    const long i = (long)(b.h0 + cy) * b.W + (b.w0 + cx);
    const long g0 = (long)n * b.P;
    float hx = 0.f, hy = 0.f;
    int r, xo, yo;
    if (tg_offsets(b, g0, i, r, xo, yo)) {
        hx = (float)xo;
        hy = (float)yo;
        if (xo < 0) hx = hx / (float)(-b.minx[g0 + r]);
        if (xo > 0) hx = hx / (float)b.maxx[g0 + r];
        if (yo < 0) hy = hy / (float)(-b.miny[g0 + r]);
        if (yo > 0) hy = hy / (float)b.maxy[g0 + r];
    }
    const long o = (long)n * b.CH * b.CW + j;
    b.hv[o * 2] = hx;
    b.hv[o * 2 + 1] = hy;
    b.np_map[o] = b.ann[g0 + i] > 0 ? 1 : 0;
}

static size_t tg_align(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t tg_carve(TgBuf &b, unsigned char *base, int n, int H, int W)
{
    const size_t NP = (size_t)n * H * W;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char *p = base ? base + off : nullptr;
        off += tg_align(bytes);
        return p;
    };
    int32_t **planes[] = {&b.par, &b.area, &b.carea, &b.rmin, &b.rmax, &b.cmin, &b.cmax, &b.comr, &b.comc, &b.minx, &b.maxx, &b.miny, &b.maxy};
    for (auto pp : planes) *pp = (int32_t *)take(NP * 4);
    b.sumr = (unsigned long long *)take(NP * 8);
    b.sumc = (unsigned long long *)take(NP * 8);
    return off;
}

size_t hvn_targets_ws_bytes(int n, int h, int w)
{
    TgBuf b;
    return tg_carve(b, nullptr, n, h, w);
}

int hvn_launch_gen_targets(const int32_t *ann, int n, int h, int w, int ch, int cw, float *hv, int32_t *np_map, void *ws, size_t ws_bytes,
                           hipStream_t stream)
{
    TgBuf b;
    if (tg_carve(b, (unsigned char *)ws, n, h, w) > ws_bytes) return -4;
    b.ann = ann; b.hv = hv; b.np_map = np_map;
    b.N = n; b.H = h; b.W = w; b.CH = ch; b.CW = cw;
    b.h0 = (int)((h - ch) * 0.5); b.w0 = (int)((w - cw) * 0.5);   // cropping_center
    b.P = (long)h * w;
    const dim3 grid((unsigned)((b.P + TG_T - 1) / TG_T), (unsigned)n);
    hipLaunchKernelGGL(tg_init, grid, dim3(TG_T), 0, stream, b);
    hipLaunchKernelGGL(tg_merge, grid, dim3(TG_T), 0, stream, b);
    hipLaunchKernelGGL(tg_stats, grid, dim3(TG_T), 0, stream, b);
    hipLaunchKernelGGL(tg_roots, grid, dim3(TG_T), 0, stream, b);
    hipLaunchKernelGGL(tg_extrema, grid, dim3(TG_T), 0, stream, b);
    const dim3 cgrid((unsigned)(((long)ch * cw + TG_T - 1) / TG_T), (unsigned)n);
    hipLaunchKernelGGL(tg_write, cgrid, dim3(TG_T), 0, stream, b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
