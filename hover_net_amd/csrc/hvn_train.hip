// hvn_train.hip -- kernels of the training step (SURVEY 8a T1/T2:
// /root/reference/models/hovernet/run_desc.py:12-109 train_step, utils.py:54-172 losses,
// torch BatchNorm2d in train() mode, torch.optim.Adam as configured by opt.py:38-44).
//
// The forward convolutions and the data gradients (dgrad = stride-1 convolution of the -- possibly dilated --
// output gradient with flipped, transposed weights) run on hvn_conv_igemm_f32 (hvn_conv.hip); this file adds
//   * hvn_conv_wgrad_f32   weight gradient as an fp32-MFMA GEMM over the pixel dimension (split-K, atomics)
//   * weight (re)packing from the parameter layout [cout][kh*kw][cin_g] to the conv kernel's slab-major layout
//     (forward) and to its transposed / tap-flipped form (dgrad), every step, on the GPU
//   * BatchNorm(train)+ReLU forward (batch statistics in double, running-stat update) and backward
//   * nearest-2x-upsample + skip-add backward, 1x1 logit head backward, conv0 weight gradient
//   * the four loss terms, forward partial sums and logit gradients (two stages around an optional all-reduce)
//   * Adam over one flat parameter slab
// All activations are NHWC fp32 strided views; every gradient destination is ACCUMULATED into (the engine zeroes
// the gradient arena once per step), see hover_net_amd/train_plan.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static inline int launch_ok() { return hipGetLastError() == hipSuccess ? 0 : -2; }

// =========================================================================================
// weight packing
// =========================================================================================
__device__ __forceinline__ void pack_w_one(const PackArgs &p, long i)
{
    const int taps = p.taps;
    if (p.mode == 2) {  // conv0: [64][7][7][3] -> [7][7][3][64] * (1/255)
        const int co = (int)(i % 64);
        const int t = (int)(i / 64);  // tap*3 + ch
        p.dst[i] = p.src[(long)co * (taps * 3) + t] * (1.0f / 255.0f);
        return;
    }
    const int l = (int)(i & 31);
    long r = i >> 5;
    const int tap = (int)(r % taps);
    r /= taps;
    const int cin = p.cin_g * p.groups;
    const int og = p.cout / p.groups;
    float v = 0.f;
    if (p.mode == 0) {  // forward: dst[co][slab][tap][l], ci = slab*32 + l
        const int slabs = cin / 32;
        const int slab = (int)(r % slabs);
        const int co = (int)(r / slabs);
        const int ci = slab * 32 + l;
        if (co < p.cout) {
            const int g = co / og;
            const int cg = ci - g * p.cin_g;
            if (cg >= 0 && cg < p.cin_g) v = p.src[((long)co * taps + tap) * p.cin_g + cg];
        }
    } else {  // dgrad: dst[ci][slab][tap'][l], co = slab*32 + l, source tap = taps-1-tap'
        const int slabs = p.cout / 32;
        const int slab = (int)(r % slabs);
        const int ci = (int)(r / slabs);
        const int co = slab * 32 + l;
        if (ci < cin) {
            const int g = co / og;
            const int cg = ci - g * p.cin_g;
            if (cg >= 0 && cg < p.cin_g) v = p.src[((long)co * taps + (taps - 1 - tap)) * p.cin_g + cg];
        }
    }
    p.dst[i] = v;
}

__device__ __forceinline__ long pack_w_total(const PackArgs &a)
{
    return a.mode == 2 ? 64L * a.taps * 3 : a.mode == 0 ? (long)a.lead_pad * (a.cin_g * a.groups) * a.taps : (long)a.lead_pad * a.cout * a.taps;
}

__global__ __launch_bounds__(256) void hvn_pack_w(const PackArgs p, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    pack_w_one(p, i);
}

// Every mode-0 / 1 / 2 packing of the step in ONE launch (a training step repacks ~260 weights: 260 launches of ~4 us were 1.1 ms of a
// 51 ms phase-1 step).  tbl[k] = the k-th packing, first_block[k] = its first workgroup (first_block[n] = the grid): a workgroup finds
// its packing by bisection.
__global__ __launch_bounds__(256) void hvn_pack_w_multi(const PackArgs *tbl, const int *first_block, int n)
{
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first_block[mid] <= b) lo = mid;
        else hi = mid - 1;
    }
    const PackArgs p = tbl[lo];
    const long i = (long)(b - first_block[lo]) * 256 + threadIdx.x;
    if (i >= pack_w_total(p)) return;
    pack_w_one(p, i);
}

int hvn_launch_pack_w_multi(const PackArgs *tbl, const int *first_block, int n, long blocks, hipStream_t stream)
{
    if (!tbl || !first_block || n <= 0 || blocks <= 0 || blocks > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(hvn_pack_w_multi, dim3((unsigned)blocks), dim3(256), 0, stream, tbl, first_block, n);
    return launch_ok();
}

// Winograd F(4x4,5x5) weight transform U[a*8+b] = sum_{r,s} G[a][r] g[r][s] G[b][s] (double accumulate), written in the
// batched-GEMM layout [64][lead_pad][k/32][32].  mode 3: forward (rows = cout, k = cin); mode 4: data gradient
// (rows = cin, k = cout, taps flipped: g'[r][s] = g[4-r][4-s]).
__global__ __launch_bounds__(256) void hvn_pack_wino(const PackArgs p, long total)
{
    // one thread = one (row, k) filter: 25 taps in, all 64 transform positions out (k fastest: coalesced stores)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int K = p.mode == 3 ? p.cin_g : p.cout;
    const int rows = p.mode == 3 ? p.cout : p.cin_g;
    const int k = (int)(i % K);
    const int row = (int)(i / K);
    const long plane = (long)p.lead_pad * K;
    float *dst = p.dst + (long)row * K + k;
    if (row >= rows) {
#pragma unroll
        for (int pos = 0; pos < 64; ++pos) dst[pos * plane] = 0.f;
        return;
    }
    const int co = p.mode == 3 ? row : k, ci = p.mode == 3 ? k : row;
    const float *g = p.src + (long)co * 25 * p.cin_g + ci;
    double gm[40];
#pragma unroll
    for (int e = 0; e < 40; ++e) gm[e] = (double)p.gmat[e];
    double tmp[8][5];   // G g
#pragma unroll
    for (int s2 = 0; s2 < 5; ++s2) {
        double col[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int tap = p.mode == 3 ? r * 5 + s2 : (4 - r) * 5 + (4 - s2);
            col[r] = (double)g[(long)tap * p.cin_g];
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 5; ++r) acc += gm[a * 5 + r] * col[r];
            tmp[a][s2] = acc;
        }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < 5; ++s2) acc += tmp[a][s2] * gm[b * 5 + s2];
            dst[(a * 8 + b) * plane] = (float)acc;
        }
}

int hvn_launch_pack_w(const PackArgs &a, hipStream_t stream)
{
    long total;
    if (a.mode == 3 || a.mode == 4) {
        if (a.groups != 1 || a.taps != 25 || !a.gmat) return -1;
        total = (long)a.lead_pad * (a.mode == 3 ? a.cin_g : a.cout);
        hipLaunchKernelGGL(hvn_pack_wino, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
        return launch_ok();
    }
    if (a.mode == 2)
        total = 64L * a.taps * 3;
    else if (a.mode == 0)
        total = (long)a.lead_pad * (a.cin_g * a.groups) * a.taps;
    else
        total = (long)a.lead_pad * a.cout * a.taps;
    hipLaunchKernelGGL(hvn_pack_w, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    return launch_ok();
}

// =========================================================================================
// weight gradient: dW[co][tap][ci] = sum_{n,oy,ox} dY[n,oy,ox,co] * X[n, oy*s+r-pt, ox*s+q-pl, ci]
// GEMM with M = cout (A = dY), N = cin (B = X shifted by the tap), K = pixels.  Both operands are K-major in
// memory (NHWC: a pixel's channels are contiguous), so a k-step of 32 pixels is staged [pixel][channel] with
// 16-byte loads/stores and the MFMA fragments are read straight out of that layout: lane l of
// v_mfma_f32_32x32x2_f32 needs A[m = l&31][k = l>>5]; reading LDS[k0 + (l>>5)][col0 + (l&31)*MB + i] for the
// wave's i-th 32-wide block means block i owns the columns congruent to i (mod MB) -- an arbitrary but fixed
// relabelling of the M index that the epilogue undoes.  One ds_read_b64 (MB = 2) feeds two MFMA blocks.
// Grid: x = (tap, cin tile, cout tile), y = K split; partial products are combined with fp32 atomics.
// =========================================================================================
template <int B, int MB>
struct WgPitch {
    static constexpr int value = (MB == 2) ? (B + 8) : ((B + 63) / 64 * 64 + 32);
};

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 2) void hvn_conv_wgrad_f32(const WgradArgs p)
{
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MBA = WM / 32, MBB = WN / 32;
    static_assert(WAVES_M * WAVES_N == 4 && (MBA == 1 || MBA == 2) && (MBB == 1 || MBB == 2), "tile shape");
    constexpr int PITCH_A = WgPitch<BM, MBA>::value, PITCH_B = WgPitch<BN, MBB>::value;
    constexpr int RPA = 1024 / BM, RPB = 1024 / BN;  // rows staged per pass (256 threads x float4)
    constexpr int PA = 32 / RPA, PB = 32 / RPB;      // passes per 32-row k-step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                     // [2][32][PITCH_A]
    float *Bs = smem + 2 * 32 * PITCH_A;  // [2][32][PITCH_B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const float *px = p.x + (long)blockIdx.z * p.xb;      // batched launch: blockIdx.z = one of nbatch independent problems
    const float *pdy = p.dy + (long)blockIdx.z * p.db;
    float *pdw = p.dw + (long)blockIdx.z * p.wb;
    int bid = blockIdx.x;
    const int tm = bid % p.tiles_m;
    bid /= p.tiles_m;
    const int tn = bid % p.tiles_n;
    const int tap = bid / p.tiles_n;
    const int tr = tap / p.KW, tq = tap - tr * p.KW;
    const int m0 = tm * BM, n0 = tn * BN;

    const unsigned R = (unsigned)p.N * p.Ho * p.Wo;
    const unsigned r_begin = blockIdx.y * p.rows_per_split;
    const unsigned r_end = min(R, r_begin + p.rows_per_split);
    if (r_begin >= r_end) return;
    const int steps = (int)((r_end - r_begin + 31) / 32);

    // per-thread staging coordinates: row (n, oy, ox) advanced incrementally by 32 per k-step
    const int a_c4 = tid % (BM / 4), a_r = tid / (BM / 4);
    const int b_c4 = tid % (BN / 4), b_r = tid / (BN / 4);
    int an[PA], ay[PA], ax[PA], bn_[PB], by[PB], bx[PB];
    unsigned arow[PA], brow[PB];
    const unsigned HoWo = (unsigned)p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const unsigned r = r_begin + a_r + RPA * j;
        arow[j] = r;
        an[j] = r / HoWo;
        const unsigned rem = r - an[j] * HoWo;
        ay[j] = rem / p.Wo;
        ax[j] = rem - ay[j] * p.Wo;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const unsigned r = r_begin + b_r + RPB * j;
        brow[j] = r;
        bn_[j] = r / HoWo;
        const unsigned rem = r - bn_[j] * HoWo;
        by[j] = rem / p.Wo;
        bx[j] = rem - by[j] * p.Wo;
    }
    auto advance = [&](int &n, int &y, int &x) {
        x += 32;
        while (x >= p.Wo) { x -= p.Wo; ++y; }     // at most a few trips: 32 pixels rarely span more than two rows
        while (y >= p.Ho) { y -= p.Ho; ++n; }
    };

    f32x4 ra[PA], rb[PB];
    auto load_global = [&]() {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (arow[j] < r_end) v = *(const f32x4 *)(pdy + (long)an[j] * p.dsn + (long)ay[j] * p.dsy + (long)ax[j] * p.dsx + m0 + a_c4 * 4);
            ra[j] = v;
            arow[j] += 32;
            advance(an[j], ay[j], ax[j]);
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int iy = by[j] * p.stride + tr - p.pad_t, ix = bx[j] * p.stride + tq - p.pad_l;
            if (brow[j] < r_end && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v = *(const f32x4 *)(px + (long)bn_[j] * p.xsn + (long)iy * p.xsy + (long)ix * p.xsx + n0 + b_c4 * 4);
            rb[j] = v;
            brow[j] += 32;
            advance(bn_[j], by[j], bx[j]);
        }
    };
    auto store_lds = [&](int buf) {
        float *a = As + buf * 32 * PITCH_A, *b = Bs + buf * 32 * PITCH_B;
#pragma unroll
        for (int j = 0; j < PA; ++j) *(f32x4 *)(a + (a_r + RPA * j) * PITCH_A + a_c4 * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < PB; ++j) *(f32x4 *)(b + (b_r + RPB * j) * PITCH_B + b_c4 * 4) = rb[j];
    };

    f32x16 acc[MBA][MBB];
#pragma unroll
    for (int i = 0; i < MBA; ++i)
#pragma unroll
        for (int j = 0; j < MBB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const float *a = As + buf * 32 * PITCH_A + lh * PITCH_A + wm * WM + l31 * MBA;
        const float *b = Bs + buf * 32 * PITCH_B + lh * PITCH_B + wn * WN + l31 * MBB;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float fa[MBA], fb[MBB];
            if constexpr (MBA == 2) {
                const f32x2 v = *(const f32x2 *)(a + 2 * t * PITCH_A);
                fa[0] = v.x;
                fa[1] = v.y;
            } else
                fa[0] = a[2 * t * PITCH_A];
            if constexpr (MBB == 2) {
                const f32x2 v = *(const f32x2 *)(b + 2 * t * PITCH_B);
                fb[0] = v.x;
                fb[1] = v.y;
            } else
                fb[0] = b[2 * t * PITCH_B];
#pragma unroll
            for (int i = 0; i < MBA; ++i)
#pragma unroll
                for (int j = 0; j < MBB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    load_global();
    store_lds(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const bool more = s + 1 < steps;
        if (more) load_global();
        compute(s & 1);
        if (more) store_lds((s + 1) & 1);
        __syncthreads();
    }

    // epilogue: D[m][n] of block (i, j): m = 8*(r/4) + 4*lh + (r%4) -> cout column m*MBA + i, n = l31 -> cin column n*MBB + j
    // deterministic form: this split's tile is STORED into copy blockIdx.y of the gradient tensor (every element of a copy has one
    // writer) and hvn_reduce_parts adds the copies in split order; otherwise fp32 atomics straight into the gradient
    const int taps = p.KH * p.KW;
    const int og = p.Cout / p.groups;
    float *pout = p.part ? p.part + (long)blockIdx.y * p.part_stride + (long)blockIdx.z * p.wb : pdw;
    const bool det = p.part != nullptr;
    if (p.groups == 1) {
        // dense launches (round 6, as hvn_wgrad_x3.hip): one 64-bit element offset per thread, 32-bit offsets from it for the lane's elements
        const int co0 = m0 + wm * WM + 4 * lh * MBA, ci0 = n0 + wn * WN + l31 * MBB;
        float *pbase = pout + ((long)co0 * taps + tap) * p.Cin_g + ci0;
        const unsigned row = (unsigned)(taps * p.Cin_g);
#pragma unroll
        for (int i = 0; i < MBA; ++i)
#pragma unroll
            for (int j = 0; j < MBB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dco = (8 * (r >> 2) + (r & 3)) * MBA + i;
                    if (co0 + dco >= p.Cout) continue;
                    float *d = pbase + (unsigned)dco * row + (unsigned)j;
                    if (det)
                        *d = acc[i][j][r];
                    else
                        unsafeAtomicAdd(d, acc[i][j][r]);
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < MBA; ++i)
#pragma unroll
        for (int j = 0; j < MBB; ++j) {
            const int ci = n0 + wn * WN + l31 * MBB + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 8 * (r >> 2) + 4 * lh + (r & 3);
                const int co = m0 + wm * WM + m * MBA + i;
                if (co >= p.Cout) continue;
                int cg = ci;
                if (p.groups > 1) {  // block-diagonal: only the co-group's own input channels exist in the parameter
                    cg = ci - (co / og) * p.Cin_g;
                    if (cg < 0 || cg >= p.Cin_g) continue;
                }
                float *d = pout + ((long)co * taps + tap) * p.Cin_g + cg;
                if (det)
                    *d = acc[i][j][r];
                else
                    unsafeAtomicAdd(d, acc[i][j][r]);
            }
        }
}

// dst[e] += part[0][e] + ... + part[S-1][e], always in the same order: four interleaved chains (chain j: s = j, j + 4, ... ascending),
// then (chain 0 + chain 1) + (chain 2 + chain 3).  One thread per (element, chain); the chains meet in LDS.
__global__ __launch_bounds__(256) void hvn_reduce_parts(float *__restrict__ dst, const float *__restrict__ part, long elems, long nparts, long stride)
{
    __shared__ float ch[4][64];
    const int el = threadIdx.x & 63, j = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + el;
    float s = 0.f;
    if (e < elems)
        for (long k = j; k < nparts; k += 4) s += part[k * stride + e];
    ch[j][el] = s;
    __syncthreads();
    if (j == 0 && e < elems) dst[e] += (ch[0][el] + ch[1][el]) + (ch[2][el] + ch[3][el]);
}

int hvn_launch_reduce_parts(float *dst, const float *part, long elems, long nparts, long stride, hipStream_t stream)
{
    if (!dst || !part || elems <= 0 || nparts <= 0 || (elems + 63) / 64 >= (1L << 31)) return -1;
    hipLaunchKernelGGL(hvn_reduce_parts, dim3((unsigned)((elems + 63) / 64)), dim3(256), 0, stream, dst, part, elems, nparts, stride);
    return launch_ok();
}

// The K split of a weight-gradient launch: ~`want` workgroups in all, at least `min_rows` pixels per split, rows a multiple of 32.
long hvn_wgrad_split(const WgradArgs &a, long tiles, unsigned *rows_per_split)
{
    // The K split trades matrix time against reduce traffic (every workgroup ends with one fp32 atomic, or one store, per element of its
    // tile).  Default: ~3 workgroups per slot of the 512 the chip holds, at least 8 k-steps per workgroup.  The best target depends on
    // the launch shape (profiles/r03_train_wgrad_sweep.txt: 512 for the encoder's few-tile launches at batch 4, >= 1024 for the
    // decoder's 5x5 launches), so TrainEngine.autotune_tiles times a few per shape and passes its choice in `want_wgs`.
    // HVN_WGRAD_WGS / HVN_WGRAD_MIN_ROWS override (tools/wgrad_sweep.py; read per launch so that one process can sweep them).
    const long R = (long)a.N * a.Ho * a.Wo;
    const int nb = a.nbatch > 1 ? a.nbatch : 1;
    const char *e_wgs = getenv("HVN_WGRAD_WGS"), *e_rows = getenv("HVN_WGRAD_MIN_ROWS");
    const long want = e_wgs ? atol(e_wgs) : (a.want_wgs > 0 ? a.want_wgs : 1536);
    const long min_rows = e_rows ? atol(e_rows) : 256;
    long ksplit = (want + tiles * nb - 1) / (tiles * nb);
    const long max_split = (R + min_rows - 1) / min_rows;
    if (ksplit > max_split) ksplit = max_split;
    if (ksplit < 1) ksplit = 1;
    long rps = (R + ksplit - 1) / ksplit;
    rps = (rps + 31) / 32 * 32;
    ksplit = (R + rps - 1) / rps;
    if (rows_per_split) *rows_per_split = (unsigned)rps;
    return ksplit;
}

static inline long wgrad_elems(const WgradArgs &a)
{
    return a.nbatch > 1 ? (long)a.nbatch * a.wb : (long)a.Cout * a.KH * a.KW * a.Cin_g;
}

static void wgrad_tile_shape(const WgradArgs &a, int *bm, int *bn)
{
    *bm = a.Cout >= 128 ? 128 : a.Cout >= 64 ? 64 : 32;
    *bn = a.Cin % 128 == 0 ? 128 : a.Cin % 64 == 0 ? 64 : 32;
}

long hvn_wgrad_part_floats(const WgradArgs &a, int x3)
{
    int bm, bn;
    if (x3 && hvn_wgrad_x3_supported(a))
        bm = bn = 128;
    else
        wgrad_tile_shape(a, &bm, &bn);
    const long tiles = (long)((a.Cout + bm - 1) / bm) * (a.Cin / bn) * a.KH * a.KW;
    const long ksplit = hvn_wgrad_split(a, tiles, nullptr);
    return ksplit > 1 ? ksplit * wgrad_elems(a) : 0;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_wgrad(WgradArgs a, hipStream_t stream)
{
    constexpr int MBA = BM / WAVES_M / 32, MBB = BN / WAVES_N / 32;
    constexpr size_t lds = (size_t)2 * 32 * (WgPitch<BM, MBA>::value + WgPitch<BN, MBB>::value) * sizeof(float);
    a.tiles_m = (a.Cout + BM - 1) / BM;
    a.tiles_n = a.Cin / BN;
    const long tiles = (long)a.tiles_m * a.tiles_n * a.KH * a.KW;
    const int nb = a.nbatch > 1 ? a.nbatch : 1;
    const long ksplit = hvn_wgrad_split(a, tiles, &a.rows_per_split);
    const long elems = wgrad_elems(a);
    if (a.part && ksplit > 1) {
        if (ksplit * elems > a.part_cap) return -4;
        a.part_stride = elems;
    } else
        a.part = nullptr;       // a single split: one writer per element, 0 + x in any order
    auto k = hvn_conv_wgrad_f32<BM, BN, WAVES_M, WAVES_N>;
    static std::atomic<unsigned long long> attr{0};
    if (hvn_max_lds_once(reinterpret_cast<const void *>(k), (int)lds, attr)) return -2;
    hipLaunchKernelGGL(k, dim3((unsigned)tiles, (unsigned)ksplit, (unsigned)nb), dim3(256), lds, stream, a);
    if (launch_ok()) return -2;
    return a.part ? hvn_launch_reduce_parts(a.dw, a.part, elems, ksplit, elems, stream) : 0;
}

int hvn_launch_wgrad(const WgradArgs &a, hipStream_t stream)
{
    if (a.Cin % 32 || a.Cout % 32 || a.groups < 1 || a.Cin_g * a.groups != a.Cin) return -1;
    int bm, bn;
    wgrad_tile_shape(a, &bm, &bn);
    if (bm == 128 && bn == 128) return launch_wgrad<128, 128, 2, 2>(a, stream);
    if (bm == 128 && bn == 64) return launch_wgrad<128, 64, 4, 1>(a, stream);
    if (bm == 128 && bn == 32) return launch_wgrad<128, 32, 4, 1>(a, stream);
    if (bm == 64 && bn == 128) return launch_wgrad<64, 128, 1, 4>(a, stream);
    if (bm == 64 && bn == 64) return launch_wgrad<64, 64, 2, 2>(a, stream);
    if (bm == 32 && bn == 128) return launch_wgrad<32, 128, 1, 4>(a, stream);
    return -1;   // (64|32) x (64|32)-only channel counts do not occur in this network
}


// =========================================================================================
// Weight gradient of a 5x5 stride-1 conv in the Winograd F(4x4,5x5) domain:
//   dM = A dY A^T per 4x4 output tile (hvn_wino_dy), dU[pos] = sum_tiles dM[pos]^T V[pos] (64 batched
//   hvn_conv_wgrad_f32 problems over the tiles, V = the forward pass's transformed input, kept), dg += G^T dU G
//   (hvn_wino_dw).  4 instead of 25 multiplies per output pixel, like the forward pass.
// =========================================================================================
template <int VW>
__global__ __launch_bounds__(256) void hvn_wino_dy(const WinoArgs p, long total)
{
    typedef float VT __attribute__((ext_vector_type(VW)));
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cvn = p.C / VW;
    const int cv = (int)(i % cvn);
    long t = i / cvn;
    const int T1 = p.ty * p.tx;
    const int tile = (int)(t % T1);
    const int n = (int)(t / T1);
    const int tyi = tile / p.tx, txi = tile - tyi * p.tx;
    float at[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) at[k] = p.mat[k];  // A^T [4][8], wave-uniform
    const float *src = p.x + (long)n * p.xsn + cv * VW;
    VT tmp[8][4];   // tmp[a][q] = sum_p AT[p][a] d[p][q]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        VT d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int yy = 4 * tyi + r, xx = 4 * txi + q;
            d[r] = (yy < p.H && xx < p.W) ? *(const VT *)(src + (long)yy * p.xsy + (long)xx * p.xsx) : (VT)(0.f);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            VT s = (VT)(0.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) s = __builtin_elementwise_fma((VT)(at[r * 8 + a]), d[r], s);
            tmp[a][q] = s;
        }
    }
    float *dst = p.y + (long)n * p.ysn + (long)tile * p.ysx + cv * VW;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            VT s = (VT)(0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) s = __builtin_elementwise_fma((VT)(at[q * 8 + b]), tmp[a][q], s);
            *(VT *)(dst + (long)(a * 8 + b) * p.ysy) = s;
        }
}

int hvn_launch_wino_dy(const WinoArgs &a, hipStream_t stream)
{
    if (a.C % 2) return -1;
    const long total = (long)a.N * a.ty * a.tx * (a.C / 2);
    hipLaunchKernelGGL(hvn_wino_dy<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    return launch_ok();
}

// dg[co][r][s][ci] += sum_{a,b} G[a][r] dU[a*8+b][co][ci] G[b][s]; one thread = one (co, ci) filter
__global__ __launch_bounds__(256) void hvn_wino_dw(const float *du, float *dg, const float *gmat, int cout, int cin, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i % cin);
    const int co = (int)(i / cin);
    const long plane = (long)cout * cin;
    double gm[40];
#pragma unroll
    for (int e = 0; e < 40; ++e) gm[e] = (double)gmat[e];
    double tmp[5][8];   // tmp[r][b] = sum_a G[a][r] dU[a][b]
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        double col[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) col[a] = (double)du[(long)(a * 8 + b) * plane + i];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < 8; ++a) acc += gm[a * 5 + r] * col[a];
            tmp[r][b] = acc;
        }
    }
    float *dst = dg + (long)co * 25 * cin + ci;
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 5; ++s2) {
            double acc = 0.0;
#pragma unroll
            for (int b = 0; b < 8; ++b) acc += tmp[r][b] * gm[b * 5 + s2];
            dst[(long)(r * 5 + s2) * cin] += (float)acc;
        }
}

int hvn_launch_wino_dw(const float *du, float *dg, const float *gmat, int cout, int cin, hipStream_t stream)
{
    const long total = (long)cout * cin;
    hipLaunchKernelGGL(hvn_wino_dw, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, du, dg, gmat, cout, cin, total);
    return launch_ok();
}

// =========================================================================================
// BatchNorm (train) + ReLU
// =========================================================================================
// Per-channel reductions over (n, y, x) of an NHWC view.  MODE 0: sum z, sum z^2.  MODE 1: sum g, sum g*xhat with
// g = da * (a > 0), xhat = (z - mean) * rstd.  A thread owns one channel quad (LQ lanes per row, 256/LQ rows per
// block pass), accumulates in double (8 rows in flight), the block combines through LDS and writes one partial per
// (row block, channel) to ws[block][2*c + {0,1}]; the finalize kernels sum the partials in a fixed order.
// The forward value before the ReLU (a product and a sum, two roundings: the library is built with -ffp-contract=off): the backward
// pass recomputes it from z and the saved scale / shift to get the ReLU mask `a > 0` instead of reading a -- the same two instructions
// on the same operands, hence the same predicate.
__device__ inline float bn_act(float z, float sc, float sh) { return z * sc + sh; }

template <int MODE>
__global__ __launch_bounds__(256) void hvn_bn_reduce(const BnArgs p)
{
    __shared__ double red[256][8];
    const int LQ = p.lq;
    const int q = blockIdx.x * LQ + (threadIdx.x % LQ);
    const int rsub = threadIdx.x / LQ, rper = 256 / LQ;
    const int CQ = p.C >> 2;
    const bool act = q < CQ;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 mean = (f32x4){0.f, 0.f, 0.f, 0.f}, rstd = mean, sc = mean, sh = mean;
    if (MODE == 1 && act) {
        sc = *(const f32x4 *)(p.save + q * 4);
        sh = *(const f32x4 *)(p.save + p.C + q * 4);
        mean = *(const f32x4 *)(p.save + 2 * p.C + q * 4);
        rstd = *(const f32x4 *)(p.save + 3 * p.C + q * 4);
    }
    const unsigned rows = (unsigned)p.N * p.H * p.W, W = p.W, H = p.H;   // < 2^31 (validated on the host)
    const unsigned rstep = gridDim.y * rper;
    constexpr int UN = 8;   // independent rows per iteration: all their loads are in flight before any is consumed
    for (unsigned r0 = blockIdx.y * rper + rsub; r0 < rows && act; r0 += UN * rstep) {
        f32x4 z[UN], da[UN];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const unsigned r = r0 + u * rstep;
            ok[u] = r < rows;
            const unsigned rr = ok[u] ? r : r0;
            const unsigned t = rr / W;
            const unsigned x = rr - t * W;
            const unsigned n = t / H;
            const unsigned y = t - n * H;
            z[u] = *(const f32x4 *)(p.z + (long)n * p.zsn + (long)y * p.zsy + (long)x * p.zsx + q * 4);
            if (MODE == 1) da[u] = *(const f32x4 *)(p.da + (long)n * p.gsn + (long)y * p.gsy + (long)x * p.gsx + q * 4);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (!ok[u]) continue;
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[e] += (double)z[u][e];
                    s[4 + e] += (double)z[u][e] * (double)z[u][e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = bn_act(z[u][e], sc[e], sh[e]) > 0.f ? da[u][e] : 0.f;      // a > 0, a recomputed (not re-read)
                    const float xh = (z[u][e] - mean[e]) * rstd[e];
                    s[e] += (double)g;
                    s[4 + e] += (double)(g * xh);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    if (rsub == 0 && act) {
        for (int k = 1; k < rper; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += red[threadIdx.x + k * LQ][e];
        // one partial per (row block, channel): no atomics, summed by the finalize kernel in a fixed order
        double *part = p.ws + (long)blockIdx.y * 2 * p.C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            part[2 * (q * 4 + e)] = s[e];
            part[2 * (q * 4 + e) + 1] = s[4 + e];
        }
    }
}

// Sum of the row-block partials of channel c: 8 lanes per channel read interleaved partials, LDS combine.
// Block = 32 channels x 8 part-lanes; returns the totals to part-lane 0.
__device__ inline bool bn_part_sums(const BnArgs &p, double &s1, double &s2, int &c)
{
    __shared__ double red[256][2];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    c = blockIdx.x * 32 + cl;
    double a = 0.0, b = 0.0;
    if (c < p.C)
#pragma unroll 4
        for (int k = pl; k < p.nparts; k += 8) {            // (unrolled: four pairs of loads in flight, the adds in the same order)
            a += p.ws[(long)k * 2 * p.C + 2 * c];
            b += p.ws[(long)k * 2 * p.C + 2 * c + 1];
        }
    red[threadIdx.x][0] = a;
    red[threadIdx.x][1] = b;
    __syncthreads();
    if (pl != 0 || c >= p.C) return false;
    for (int k = 1; k < 8; ++k) {
        a += red[cl + 32 * k][0];
        b += red[cl + 32 * k][1];
    }
    s1 = a;
    s2 = b;
    return true;
}

// forward finalize: batch mean / biased variance -> scale, shift, mean, rstd; running stats (momentum, unbiased var)
__global__ __launch_bounds__(256) void hvn_bn_final(const BnArgs p)
{
    double s1, s2;
    int c;
    if (!bn_part_sums(p, s1, s2, c)) return;
    const double n = (double)p.N * p.H * p.W;
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)p.eps);
    const double sc = (double)p.gamma[c] * rstd;
    p.save[c] = (float)sc;
    p.save[p.C + c] = (float)((double)p.beta[c] - mean * sc);
    p.save[2 * p.C + c] = (float)mean;
    p.save[3 * p.C + c] = (float)rstd;
    const float mom = p.momentum;
    p.running_mean[c] = (1.f - mom) * p.running_mean[c] + mom * (float)mean;
    p.running_var[c] = (1.f - mom) * p.running_var[c] + mom * (float)(var * n / (n - 1.0));
}

// backward finalize: dgamma += sum g*xhat, dbeta += sum g; coefficients of the dz formula
__global__ __launch_bounds__(256) void hvn_bn_bwd_final(const BnArgs p)
{
    double s1, s2;
    int c;
    if (!bn_part_sums(p, s1, s2, c)) return;
    const double n = (double)p.N * p.H * p.W;
    p.dgamma[c] += (float)s2;
    p.dbeta[c] += (float)s1;
    p.coef[c] = p.gamma[c] * p.save[3 * p.C + c];
    p.coef[p.C + c] = (float)(s1 / n);
    p.coef[2 * p.C + c] = (float)(s2 / n);
}

// MODE 0: a = relu(z*scale + shift).  MODE 1: dz (+)= c1 * (g - c2 - xhat*c3).
template <int MODE, bool STORE = false>
__global__ __launch_bounds__(256) void hvn_bn_apply(const BnArgs p, long total)
{
    const unsigned CQ = p.C >> 2, W = p.W, H = p.H;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {   // total < 2^31 (host)
        unsigned t = i / CQ;
        const unsigned q = i - t * CQ;
        const unsigned t2 = t / W;
        const unsigned x = t - t2 * W;
        const unsigned n = t2 / H;
        const unsigned y = t2 - n * H;
        const f32x4 z = *(const f32x4 *)(p.z + (long)n * p.zsn + (long)y * p.zsy + (long)x * p.zsx + q * 4);
        if (MODE == 0) {
            const f32x4 sc = *(const f32x4 *)(p.save + q * 4), sh = *(const f32x4 *)(p.save + p.C + q * 4);
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = fmaxf(bn_act(z[e], sc[e], sh[e]), 0.f);
            *(f32x4 *)(p.a_out + (long)n * p.asn + (long)y * p.asy + (long)x * p.asx + q * 4) = a;
        } else {
            const f32x4 da = *(const f32x4 *)(p.da + (long)n * p.gsn + (long)y * p.gsy + (long)x * p.gsx + q * 4);
            const f32x4 sc = *(const f32x4 *)(p.save + q * 4), sh = *(const f32x4 *)(p.save + p.C + q * 4);
            const f32x4 mean = *(const f32x4 *)(p.save + 2 * p.C + q * 4), rstd = *(const f32x4 *)(p.save + 3 * p.C + q * 4);
            const f32x4 c1 = *(const f32x4 *)(p.coef + q * 4), c2 = *(const f32x4 *)(p.coef + p.C + q * 4),
                        c3 = *(const f32x4 *)(p.coef + 2 * p.C + q * 4);
            float *dzp = p.dz + (long)n * p.dsn + (long)y * p.dsy + (long)x * p.dsx + q * 4;
            f32x4 dz = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!STORE) dz = *(const f32x4 *)dzp;        // STORE: this launch is the first writer of dz in the step (train_plan first-writer rule)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = bn_act(z[e], sc[e], sh[e]) > 0.f ? da[e] : 0.f;
                const float xh = (z[e] - mean[e]) * rstd[e];
                dz[e] += c1[e] * (g - c2[e] - xh * c3[e]);
            }
            *(f32x4 *)dzp = dz;
        }
    }
}

static void bn_grid(BnArgs &a, dim3 &grid)
{
    const int cq = a.C / 4;
    int lq = 64;
    while (lq > cq) lq >>= 1;      // largest power of two <= min(64, cq)
    if (lq < 1) lq = 1;
    const long rows = (long)a.N * a.H * a.W;
    const int rper = 256 / lq;
    const int gx = (cq + lq - 1) / lq;
    long gy = (rows + rper * 16 - 1) / (rper * 16);   // >= 16 rows (two 8-row iterations) per thread
    long cap = 1024 / gx;                               // ~1024 workgroups: bandwidth comes from the 8 loads in flight
    if (cap > HVN_BN_MAX_PARTS) cap = HVN_BN_MAX_PARTS;
    if (cap < 1) cap = 1;
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    a.lq = lq;
    a.nparts = (int)gy;
    grid = dim3((unsigned)gx, (unsigned)gy);
}

int hvn_launch_bn_forward(BnArgs a, hipStream_t stream)
{
    if (a.C % 4 || (long)a.N * a.H * a.W * (a.C / 4) >= (1L << 31)) return -1;
    dim3 grid;
    bn_grid(a, grid);
    hipLaunchKernelGGL(hvn_bn_reduce<0>, grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(hvn_bn_final, dim3((a.C + 31) / 32), dim3(256), 0, stream, a);
    const long total = (long)a.N * a.H * a.W * (a.C / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL((hvn_bn_apply<0, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a, total);
    return launch_ok();
}

int hvn_launch_bn_backward(BnArgs a, hipStream_t stream)
{
    if (a.C % 4 || (long)a.N * a.H * a.W * (a.C / 4) >= (1L << 31)) return -1;
    dim3 grid;
    bn_grid(a, grid);
    hipLaunchKernelGGL(hvn_bn_reduce<1>, grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(hvn_bn_bwd_final, dim3((a.C + 31) / 32), dim3(256), 0, stream, a);
    if (a.dz) {
        const long total = (long)a.N * a.H * a.W * (a.C / 4);
        long blocks = (total + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        if (a.dz_store)
            hipLaunchKernelGGL((hvn_bn_apply<1, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a, total);
        else
            hipLaunchKernelGGL((hvn_bn_apply<1, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a, total);
    }
    return launch_ok();
}

// =========================================================================================
// y = nearest2x(lo) + skip, backward: dlo += 2x2 sum of dy, dskip += dy
// =========================================================================================
__global__ __launch_bounds__(256) void hvn_upadd_bwd(const UpAddBwdArgs p, long total)
{
    const unsigned CQ = p.C >> 2, W2 = p.W / 2, H2 = p.H / 2;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
        const unsigned t = i / CQ;
        const unsigned q = i - t * CQ;
        const unsigned t2 = t / W2;
        const unsigned x = t - t2 * W2;
        const unsigned n = t2 / H2;
        const unsigned y = t2 - n * H2;
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const f32x4 g = *(const f32x4 *)(p.dy + (long)n * p.ysn + (long)(2 * y + dy) * p.ysy + (long)(2 * x + dx) * p.ysx + q * 4);
                s += g;
                if (p.dskip) {
                    float *d = p.dskip + (long)n * p.ssn + (long)(2 * y + dy) * p.ssy + (long)(2 * x + dx) * p.ssx + q * 4;
                    *(f32x4 *)d = *(const f32x4 *)d + g;
                }
            }
        if (p.dlo) {
            float *d = p.dlo + (long)n * p.lsn + (long)y * p.lsy + (long)x * p.lsx + q * 4;
            *(f32x4 *)d = *(const f32x4 *)d + s;
        }
    }
}

int hvn_launch_upadd_bwd(const UpAddBwdArgs &a, hipStream_t stream)
{
    if (a.C % 4 || a.H % 2 || a.W % 2) return -1;
    const long total = (long)a.N * (a.H / 2) * (a.W / 2) * (a.C / 4);
    if (total >= (1L << 31)) return -1;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(hvn_upadd_bwd, dim3((unsigned)blocks), dim3(256), 0, stream, a, total);
    return launch_ok();
}

// =========================================================================================
// 1x1 logit head (64 -> C, bias) backward: da += W^T dl, dW += sum dl (x) a, db += sum dl
// =========================================================================================
// Every lane owns one pixel; the weight / bias gradients are summed across the wave with a butterfly (6 exchanges: the same tree every
// run) into the wave's OWN row of an LDS table -- a plain read-modify-write by the wave's first lane, no LDS atomics (round 3 added every
// lane's product with its own LDS atomic: 64 lanes on one address, 0.36 ms per head; rounds 4-5 one atomic per wave and value, whose order
// among the four waves was not fixed).  A workgroup walks several 256-pixel groups (grid capped at HEAD_BWD_MAX_WGS), then sums its four
// rows in wave order and leaves with one fp32 atomic per value -- or, deterministic form, STORES its sums as row blockIdx.x of `part`,
// which hvn_reduce_parts adds to dW / db in workgroup order.
#define HEAD_BWD_MAX_WGS 512
__device__ __forceinline__ float head_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(256) void hvn_head_bwd(const HeadBwdArgs p, long total)
{
    __shared__ float sw[4][16 * 64 + 16];  // per wave: dW tile then db
    const int C = p.Cout;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * (16 * 64 + 16); i += 256) (&sw[0][0])[i] = 0.f;
    __syncthreads();
    float *mine = sw[wave];
    const bool lead = (threadIdx.x & 63) == 0;
    const long plane = (long)p.H * p.W;
    for (long base = (long)blockIdx.x * 256; base < total; base += (long)gridDim.x * 256) {
        const long i = base + threadIdx.x;
        const bool act = i < total;
        const long ii = act ? i : 0;
        const int x = (int)(ii % p.W);
        long t = ii / p.W;
        const int y = (int)(t % p.H);
        const int n = (int)(t / p.H);
        const float *src = p.x + (long)n * p.xsn + (long)y * p.xsy + (long)x * p.xsx;
        float *dst = p.dx + (long)n * p.dsn + (long)y * p.dsy + (long)x * p.dsx;
        const float *dl = p.dl + (long)n * C * plane + (long)y * p.W + x;
        float a[64], da[64];
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
            const f32x4 v = *(const f32x4 *)(src + c);
            const f32x4 d = *(const f32x4 *)(dst + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[c + e] = act ? v[e] : 0.f;
                da[c + e] = d[e];
            }
        }
        for (int co = 0; co < C; ++co) {
            const float g = act ? dl[co * plane] : 0.f;
            const float *__restrict__ w = p.w + co * 64;
            const float gs = head_wave_sum(g);
            if (lead) mine[C * 64 + co] += gs;
#pragma unroll
            for (int c = 0; c < 64; ++c) {
                da[c] = fmaf(g, w[c], da[c]);
                const float ws = head_wave_sum(g * a[c]);
                if (lead) mine[co * 64 + c] += ws;
            }
        }
        if (act) {
#pragma unroll
            for (int c = 0; c < 64; c += 4) *(f32x4 *)(dst + c) = (f32x4){da[c], da[c + 1], da[c + 2], da[c + 3]};
        }
    }
    __syncthreads();
    const int nv = C * 64 + C;
    for (int k = threadIdx.x; k < nv; k += 256) {
        const float v = (sw[0][k] + sw[1][k]) + (sw[2][k] + sw[3][k]);
        if (p.part)
            p.part[(long)blockIdx.x * nv + k] = v;
        else
            unsafeAtomicAdd(k < C * 64 ? p.dw + k : p.db + (k - C * 64), v);
    }
}

static inline long head_bwd_wgs(const HeadBwdArgs &a)
{
    const long groups = ((long)a.N * a.H * a.W + 255) / 256;
    return groups < HEAD_BWD_MAX_WGS ? groups : HEAD_BWD_MAX_WGS;
}
long hvn_head_bwd_part_floats(const HeadBwdArgs &a) { return head_bwd_wgs(a) * (a.Cout * 64 + a.Cout); }

int hvn_launch_head_bwd(const HeadBwdArgs &a, hipStream_t stream)
{
    if (a.Cout < 1 || a.Cout > 16) return -1;
    const long total = (long)a.N * a.H * a.W;
    const long wgs = head_bwd_wgs(a);
    const int nv = a.Cout * 64 + a.Cout;
    if (a.part && wgs * nv > a.part_cap) return -4;
    hipLaunchKernelGGL(hvn_head_bwd, dim3((unsigned)wgs), dim3(256), 0, stream, a, total);
    if (launch_ok()) return -2;
    if (!a.part) return 0;
    if (hvn_launch_reduce_parts(a.dw, a.part, (long)a.Cout * 64, wgs, nv, stream)) return -2;
    return hvn_launch_reduce_parts(a.db, a.part + (long)a.Cout * 64, a.Cout, wgs, nv, stream);
}

// =========================================================================================
// conv0 weight gradient: dW[co][r][s][ch] += sum dz[n,oy,ox,co] * img[n, oy+r-pad, ox+s-pad, ch] / 255
// One workgroup walks several 16x16 output tiles: image patch (22x22x3) and the dz tile (256 px x 64 co) in LDS;
// thread = (co, tap phase): 37 of the 147 (r, s*3+ch) taps each, register accumulators across the tiles.
// =========================================================================================
#define W0_T 16
#define W0_P (W0_T + 6)
__global__ __launch_bounds__(256) void hvn_conv0_wgrad(const Conv0WgradArgs p, int tiles_x, int tiles_y, long tiles_total)
{
    __shared__ float patch[W0_P][W0_P * 3 + 2];
    __shared__ float dz[W0_T * W0_T / 2][64];
    const int tid = threadIdx.x;
    const int co = tid & 63;
    const int ph = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform tap phase
    float acc[37];
#pragma unroll
    for (int k = 0; k < 37; ++k) acc[k] = 0.f;
    for (long tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x);
        const long t2 = tile / tiles_x;
        const int ty = (int)(t2 % tiles_y);
        const int n = (int)(t2 / tiles_y);
        const int oy0 = ty * W0_T, ox0 = tx * W0_T;
        __syncthreads();
        const uint8_t *img = p.img + (long)n * p.isn;
        for (int i = tid; i < W0_P * W0_P * 3; i += 256) {
            const int py = i / (W0_P * 3), pr = i - py * (W0_P * 3);
            const int px = pr / 3, ch = pr - px * 3;
            const int iy = oy0 + py - p.pad, ix = ox0 + px - p.pad;
            float v = 0.f;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = (float)img[(long)iy * p.isy + (long)ix * p.isx + ch];
            patch[py][pr] = v * (1.0f / 255.0f);
        }
        for (int half = 0; half < 2; ++half) {      // the dz tile goes through LDS in two halves of 8 rows
            if (half) __syncthreads();
            for (int i = tid; i < (W0_T * W0_T / 2) * 16; i += 256) {
                const int pix = i >> 4, c4 = i & 15;
                const int oy = oy0 + half * 8 + (pix >> 4), ox = ox0 + (pix & 15);
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (oy < p.Ho && ox < p.Wo) v = *(const f32x4 *)(p.dy + (long)n * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + c4 * 4);
                *(f32x4 *)&dz[pix][c4 * 4] = v;
            }
            __syncthreads();
            for (int pix = 0; pix < W0_T * W0_T / 2; ++pix) {
                const float g = dz[pix][co];
                const float *prow = &patch[half * 8 + (pix >> 4)][(pix & 15) * 3];
#pragma unroll
                for (int k = 0; k < 37; ++k) {
                    const int tap = ph + 4 * k;  // 0..146 (the last phases have one tap less)
                    if (tap < 147) {
                        const int r = tap / 21, s3 = tap - r * 21;
                        acc[k] = fmaf(g, prow[r * (W0_P * 3 + 2) + s3], acc[k]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 37; ++k) {
        const int tap = ph + 4 * k;
        if (tap >= 147) continue;
        if (p.part)
            p.part[(long)blockIdx.x * (64 * 147) + co * 147 + tap] = acc[k];
        else
            unsafeAtomicAdd(p.dw + (long)co * 147 + tap, acc[k]);
    }
}

// conv0 weight gradient on the matrix cores: D[co][kidx] = sum_px dz[px][co] * P[px][kidx] with P gathered from the staged
// image patch (kidx -> (tap row, tap col*3 + ch) is a per-lane constant offset), M = 64 channels (2 blocks), N = 147 (+13)
// taps (5 blocks), K = pixels.  A wave reduces 32 pixels of each 8-row half tile (16 k-steps x 10 MFMAs) and keeps its
// 64 x 160 partial in registers across all the tiles of the workgroup; the four waves' partials are summed through LDS and
// leave with one atomic per weight.  ~10x the VALU kernel above (HVN_CONV0_WGRAD_VALU=1 keeps that one).
#define W0M_PITCH (W0_P * 3 + 2)
__global__ __launch_bounds__(256) void hvn_conv0_wgrad_mfma(const Conv0WgradArgs p, int tiles_x, int tiles_y, long tiles_total)
{
    typedef float f32x16v __attribute__((ext_vector_type(16)));
    extern __shared__ __attribute__((aligned(16))) float w0m_lds[];
    float *patch = w0m_lds;                              // [W0_P][W0M_PITCH]
    float *dzs = w0m_lds + W0_P * W0M_PITCH;             // [128][64]; later the [64][160] reduction buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16v acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int boff[5];
    float bmask[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int kidx = 32 * j + l31;
        const bool ok = kidx < 147;
        boff[j] = ok ? (kidx / 21) * W0M_PITCH + (kidx % 21) : 0;
        bmask[j] = ok ? 1.f : 0.f;
    }
    for (long tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x);
        const long t2 = tile / tiles_x;
        const int ty = (int)(t2 % tiles_y);
        const int n = (int)(t2 / tiles_y);
        const int oy0 = ty * W0_T, ox0 = tx * W0_T;
        __syncthreads();
        const uint8_t *img = p.img + (long)n * p.isn;
        for (int i = tid; i < W0_P * W0_P * 3; i += 256) {
            const int py = i / (W0_P * 3), pr = i - py * (W0_P * 3);
            const int px = pr / 3, ch = pr - px * 3;
            const int iy = oy0 + py - p.pad, ix = ox0 + px - p.pad;
            float v = 0.f;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = (float)img[(long)iy * p.isy + (long)ix * p.isx + ch];
            patch[py * W0M_PITCH + pr] = v * (1.0f / 255.0f);
        }
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
            for (int i = tid; i < 128 * 16; i += 256) {
                const int pix = i >> 4, c4 = i & 15;
                const int oy = oy0 + half * 8 + (pix >> 4), ox = ox0 + (pix & 15);
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (oy < p.Ho && ox < p.Wo) v = *(const f32x4 *)(p.dy + (long)n * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + c4 * 4);
                *(f32x4 *)(dzs + pix * 64 + c4 * 4) = v;
            }
            __syncthreads();
            // this wave: rows 2*wave, 2*wave+1 of the half tile = pixels 32*wave .. 32*wave+31; k-step t: pixel 2t + lh
            const float *abase = dzs + (32 * wave + lh) * 64 + l31;
            const float *pbase = patch + (half * 8 + 2 * wave) * W0M_PITCH + lh * 3;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int prow = t >> 3, pcol = 2 * (t & 7);             // pixel 2t: row, column (lh adds one column)
                const float a0 = abase[2 * t * 64], a1 = abase[2 * t * 64 + 32];
                const float *pb = pbase + prow * W0M_PITCH + pcol * 3;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const float b = pb[boff[j]] * bmask[j];
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][j], 0, 0, 0);
                }
            }
        }
    }
    // sum the four waves' partials in LDS ([64 co][160 taps]) in WAVE ORDER (four turns; LDS atomics left the order to the scheduler), then
    // one atomic per weight -- or, deterministic form, the workgroup's sums stored as row blockIdx.x of `part` for hvn_reduce_parts
    __syncthreads();
    float *red = dzs;
    for (int i = tid; i < 64 * 160; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        red[co * 160 + 32 * j + l31] += acc[i][j][r];
                    }
        }
        __syncthreads();
    }
    for (int i = tid; i < 64 * 147; i += 256) {
        const int co = i / 147, k = i - co * 147;
        if (p.part)
            p.part[(long)blockIdx.x * (64 * 147) + i] = red[co * 160 + k];
        else
            unsafeAtomicAdd(p.dw + i, red[co * 160 + k]);
    }
}

static inline bool conv0_wgrad_valu()
{
    static int valu = -1;
    if (valu < 0) valu = getenv("HVN_CONV0_WGRAD_VALU") ? 1 : 0;
    return valu != 0;
}
static inline long conv0_wgrad_wgs(const Conv0WgradArgs &a)
{
    const int tx = (a.Wo + W0_T - 1) / W0_T, ty = (a.Ho + W0_T - 1) / W0_T;
    const long total = (long)tx * ty * a.N, cap = conv0_wgrad_valu() ? 1024 : 512;
    return total < cap ? total : cap;
}
long hvn_conv0_wgrad_part_floats(const Conv0WgradArgs &a) { return conv0_wgrad_wgs(a) * 64 * 147; }

int hvn_launch_conv0_wgrad(const Conv0WgradArgs &a, hipStream_t stream)
{
    const int tx = (a.Wo + W0_T - 1) / W0_T, ty = (a.Ho + W0_T - 1) / W0_T;
    const long total = (long)tx * ty * a.N;
    const long blocks = conv0_wgrad_wgs(a);
    if (a.part && blocks * 64 * 147 > a.part_cap) return -4;
    if (!conv0_wgrad_valu()) {
        const size_t lds = (size_t)(W0_P * W0M_PITCH + 64 * 160) * sizeof(float);
        static std::atomic<unsigned long long> attr{0};
        if (hvn_max_lds_once(reinterpret_cast<const void *>(hvn_conv0_wgrad_mfma), (int)lds, attr)) return -2;
        hipLaunchKernelGGL(hvn_conv0_wgrad_mfma, dim3((unsigned)blocks), dim3(256), lds, stream, a, tx, ty, total);
    } else
        hipLaunchKernelGGL(hvn_conv0_wgrad, dim3((unsigned)blocks), dim3(256), 0, stream, a, tx, ty, total);
    if (launch_ok()) return -2;
    return a.part ? hvn_launch_reduce_parts(a.dw, a.part, 64 * 147, blocks, 64 * 147, stream) : 0;
}

// =========================================================================================
// losses (utils.py:54-172 as composed by run_desc.py:40-82 with the weight table of opt.py:47-51: wt[] scales the gradients)
//   sums[0] bce_np  [1] bce_tp  [2] mse  [3] msge numerator  [4] focus sum (both channels)
//   sums[8 + c]  dice np: inse[c], [10 + c] l[c], [12 + c] r[c]   (c < 2)
//   sums[16 + c] dice tp: inse[c], [32 + c] l[c], [48 + c] r[c]   (c < T <= 16)
// Stage 1 accumulates the sums of this rank's pixels and stores the focus-weighted Sobel differences; the host
// may all-reduce `sums` over the ranks; stage 2 turns the (global) sums into the logit gradients of this rank's
// pixels, so that SUM-reducing the parameter gradients over ranks gives the full-batch gradient of the reference's
// single-process DataParallel step.
// =========================================================================================
__device__ inline float sobel5_h(int r, int s)  // kernel_h[r][s] = h / (h^2 + v^2 + 1e-15), h = r-2, v = s-2 (utils.py:135-137)
{
    const float h = (float)(r - 2), v = (float)(s - 2);
    return h / (h * h + v * v + 1.0e-15f);
}

// Reduction of the sums, the same tree every run (rounds 1-5: double atomics in LDS and into `sums`, whose order the scheduler picked -- a
// difference in the 16th digit, but one that a float conversion downstream can turn into a different gradient bit): every value is summed
// across the wave by a butterfly, lands in the wave's own LDS row, the four rows are added in wave order, and the workgroup's 64 sums
// leave with one double atomic each -- or, deterministic form (p.parts), are STORED as row blockIdx.x of `parts`, which
// hvn_loss_finalize adds to `sums` in a fixed order.
__device__ __forceinline__ double loss_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void hvn_loss_partial(const LossArgs p, long total)
{
    __shared__ double red[4][64];
    (&red[0][0])[threadIdx.x] = 0.0;
    __syncthreads();
    double *mine = red[threadIdx.x >> 6];
    const bool lead = (threadIdx.x & 63) == 0;
    auto put = [&](int slot, double v) {          // wave-uniform control flow around every call
        const double s = loss_wave_sum(v);
        if (lead) mine[slot] = s;
    };
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool act = i < total;
    const long ii = act ? i : 0;
    const double on = act ? 1.0 : 0.0;
    {
        const int x = (int)(ii % p.W);
        long t = ii / p.W;
        const int y = (int)(t % p.H);
        const int n = (int)(t / p.H);
        const long plane = (long)p.H * p.W;
        const long pix = (long)y * p.W + x;
        const float eps = 10e-8f;
        // np branch
        {
            const float *l = p.l_np + (long)n * 2 * plane + pix;
            const float l0 = l[0], l1 = l[plane];
            const float mx = fmaxf(l0, l1);
            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
            const float inv = 1.f / (e0 + e1);
            const float pr[2] = {e0 * inv, e1 * inv};
            const int tc = p.t_np[ii] != 0;
            const float q = fminf(fmaxf(pr[tc] / (pr[0] + pr[1]), eps), 1.f - eps);
            put(0, on * (double)(-logf(q)));
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                put(8 + c, c == tc ? on * (double)pr[c] : 0.0);
                put(10 + c, on * (double)pr[c]);
                put(12 + c, c == tc ? on : 0.0);
            }
        }
        // tp branch
        if (p.T > 0) {
            const float *l = p.l_tp + (long)n * p.T * plane + pix;
            float mx = l[0];
            for (int c = 1; c < p.T; ++c) mx = fmaxf(mx, l[c * plane]);
            float pr[16], sum = 0.f;
            for (int c = 0; c < p.T; ++c) {
                pr[c] = expf(l[c * plane] - mx);
                sum += pr[c];
            }
            const float inv = 1.f / sum;
            float ps = 0.f;
            for (int c = 0; c < p.T; ++c) {
                pr[c] *= inv;
                ps += pr[c];
            }
            const int tc = p.t_tp[ii];
            float ptc = pr[0];
            for (int c = 1; c < p.T; ++c) ptc = c == tc ? pr[c] : ptc;
            const float q = fminf(fmaxf(ptc / ps, eps), 1.f - eps);
            put(1, on * (double)(-logf(q)));
            for (int c = 0; c < p.T; ++c) {
                put(16 + c, c == tc ? on * (double)pr[c] : 0.0);
                put(32 + c, on * (double)pr[c]);
                put(48 + c, c == tc ? on : 0.0);
            }
        }
        // hv branch: mse + msge (Sobel of the difference, zero padding 2)
        {
            const float *l = p.l_hv + (long)n * 2 * plane;
            const float *tv = p.t_hv + (long)n * plane * 2;
            const float d0 = l[pix] - tv[pix * 2], d1 = l[plane + pix] - tv[pix * 2 + 1];
            put(2, on * (double)(d0 * d0 + d1 * d1));
            float g0 = 0.f, g1 = 0.f;
            for (int r = 0; r < 5; ++r)
                for (int s = 0; s < 5; ++s) {
                    const int yy = y + r - 2, xx = x + s - 2;
                    if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) {
                        const long q2 = (long)yy * p.W + xx;
                        g0 = fmaf(sobel5_h(r, s), l[q2] - tv[q2 * 2], g0);
                        g1 = fmaf(sobel5_h(s, r), l[plane + q2] - tv[q2 * 2 + 1], g1);
                    }
                }
            const float f = p.t_np[ii] != 0 ? 1.f : 0.f;
            put(3, on * (double)(f * (g0 * g0 + g1 * g1)));
            put(4, on * (double)(2.f * f));
            if (act) {
                p.gws[i * 2] = f * g0;
                p.gws[i * 2 + 1] = f * g1;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int k = threadIdx.x;
        const double v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        if (p.parts)
            p.parts[(long)blockIdx.x * 64 + k] = v;
        else if (v != 0.0)
            unsafeAtomicAdd(p.sums + k, v);
    }
}

// sums[k] += parts[0][k] + parts[1][k] + ... in a fixed order: 16 interleaved chains per value (chain j: rows j, j + 16, ... ascending),
// folded pairwise in LDS.  One workgroup of 64 values x 16 chains.
__global__ __launch_bounds__(1024) void hvn_loss_finalize(const double *__restrict__ parts, long rows, double *__restrict__ sums)
{
    __shared__ double ch[16][64];
    const int k = threadIdx.x & 63, j = threadIdx.x >> 6;
    double s = 0.0;
    for (long r = j; r < rows; r += 16) s += parts[r * 64 + k];
    ch[j][k] = s;
    __syncthreads();
    for (int w = 8; w > 0; w >>= 1) {
        if (j < w) ch[j][k] += ch[j + w][k];
        __syncthreads();
    }
    if (j == 0) sums[k] += ch[0][k];
}

long hvn_loss_part_doubles(int n, int h, int w) { return (((long)n * h * w + 255) / 256) * 64; }

__global__ __launch_bounds__(256) void hvn_loss_grad(const LossArgs p, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % p.W);
    long t = i / p.W;
    const int y = (int)(t % p.H);
    const int n = (int)(t / p.H);
    const long plane = (long)p.H * p.W;
    const long pix = (long)y * p.W + x;
    const float eps = 10e-8f, smooth = 1e-3f;
    const double M = p.m_total;  // pixels of the full (all-rank) batch
    {
        const float *l = p.l_np + (long)n * 2 * plane + pix;
        const float l0 = l[0], l1 = l[plane];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float inv = 1.f / (e0 + e1);
        const float pr[2] = {e0 * inv, e1 * inv};
        const int tc = p.t_np[i] != 0;
        const float q = pr[tc] / (pr[0] + pr[1]);
        const float gate = (q > eps && q < 1.f - eps) ? 1.f : 0.f;
        double D[2], dot = 0.0;     // the dice term cancels heavily (D[c] - sum_j D[j] p[j]): combine in double
        for (int c = 0; c < 2; ++c) {
            const double I = p.sums[8 + c], den = p.sums[10 + c] + p.sums[12 + c] + smooth;
            const double tcf = c == tc ? 1.0 : 0.0;
            D[c] = -(2.0 * tcf * den - (2.0 * I + smooth)) / (den * den);
            dot += D[c] * (double)pr[c];
        }
        float *d = p.d_np + (long)n * 2 * plane + pix;
        for (int c = 0; c < 2; ++c)
            d[c * plane] = (float)((double)p.wt[0] * ((double)gate * ((double)pr[c] - (c == tc ? 1.0 : 0.0)) / M) +
                                   (double)p.wt[1] * ((double)pr[c] * (D[c] - dot)));
    }
    if (p.T > 0) {
        const float *l = p.l_tp + (long)n * p.T * plane + pix;
        float mx = l[0];
        for (int c = 1; c < p.T; ++c) mx = fmaxf(mx, l[c * plane]);
        float pr[16], sum = 0.f;
        for (int c = 0; c < p.T; ++c) {
            pr[c] = expf(l[c * plane] - mx);
            sum += pr[c];
        }
        const float inv = 1.f / sum;
        float ps = 0.f;
        for (int c = 0; c < p.T; ++c) {
            pr[c] *= inv;
            ps += pr[c];
        }
        const int tc = p.t_tp[i];
        const float q = pr[tc] / ps;
        const float gate = (q > eps && q < 1.f - eps) ? 1.f : 0.f;
        double D[16], dot = 0.0;
        for (int c = 0; c < p.T; ++c) {
            const double I = p.sums[16 + c], den = p.sums[32 + c] + p.sums[48 + c] + smooth;
            const double tcf = c == tc ? 1.0 : 0.0;
            D[c] = -(2.0 * tcf * den - (2.0 * I + smooth)) / (den * den);
            dot += D[c] * (double)pr[c];
        }
        float *d = p.d_tp + (long)n * p.T * plane + pix;
        for (int c = 0; c < p.T; ++c)
            d[c * plane] = (float)((double)p.wt[4] * ((double)gate * ((double)pr[c] - (c == tc ? 1.0 : 0.0)) / M) +
                                   (double)p.wt[5] * ((double)pr[c] * (D[c] - dot)));
    }
    {
        const float *l = p.l_hv + (long)n * 2 * plane;
        const float *tv = p.t_hv + (long)n * plane * 2;
        const float d0 = l[pix] - tv[pix * 2], d1 = l[plane + pix] - tv[pix * 2 + 1];
        // msge: dL/de[y,x] = sum_{r,s} K[r][s] * 2 f G[y-r+2, x-s+2] / (F + 1e-8)
        float a0 = 0.f, a1 = 0.f;
        const float *gw = p.gws + (long)n * plane * 2;
        for (int r = 0; r < 5; ++r)
            for (int s = 0; s < 5; ++s) {
                const int yy = y - r + 2, xx = x - s + 2;
                if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) {
                    const long q2 = ((long)yy * p.W + xx) * 2;
                    a0 = fmaf(sobel5_h(r, s), gw[q2], a0);
                    a1 = fmaf(sobel5_h(s, r), gw[q2 + 1], a1);
                }
            }
        const float kf = (float)(2.0 / (p.sums[4] + 1.0e-8));
        float *d = p.d_hv + (long)n * 2 * plane + pix;
        d[0] = p.wt[2] * (d0 / (float)M) + p.wt[3] * (kf * a0);        // mse: mean over M*2 elements of e^2 -> e / M
        d[plane] = p.wt[2] * (d1 / (float)M) + p.wt[3] * (kf * a1);
    }
}

int hvn_launch_loss(const LossArgs &a, int stage, hipStream_t stream)
{
    if (a.T < 0 || a.T > 16) return -1;
    const long total = (long)a.N * a.H * a.W;
    if (stage == 0) {
        const long rows = (total + 255) / 256;
        if (a.parts && rows * 64 > a.parts_cap) return -4;
        hipLaunchKernelGGL(hvn_loss_partial, dim3((unsigned)rows), dim3(256), 0, stream, a, total);
        if (a.parts) {
            if (launch_ok()) return -2;
            hipLaunchKernelGGL(hvn_loss_finalize, dim3(1), dim3(1024), 0, stream, (const double *)a.parts, rows, a.sums);
        }
    } else
        hipLaunchKernelGGL(hvn_loss_grad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    return launch_ok();
}

// =========================================================================================
// Adam (torch.optim.Adam, no weight decay / amsgrad) over one flat slab
// =========================================================================================
__global__ __launch_bounds__(256) void hvn_adam(float *w, const float *g, float *m, float *v, long n, float b1, float b2, float eps,
                                                float step_size, float inv_bc2_sqrt)
{
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const f32x4 gg = *(const f32x4 *)(g + i);
            f32x4 mm = *(const f32x4 *)(m + i), vv = *(const f32x4 *)(v + i), ww = *(const f32x4 *)(w + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mm[e] = b1 * mm[e] + (1.f - b1) * gg[e];
                vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
                ww[e] -= step_size * mm[e] / (sqrtf(vv[e]) * inv_bc2_sqrt + eps);
            }
            *(f32x4 *)(m + i) = mm;
            *(f32x4 *)(v + i) = vv;
            *(f32x4 *)(w + i) = ww;
        } else {
            for (long k = i; k < n; ++k) {
                const float gk = g[k];
                m[k] = b1 * m[k] + (1.f - b1) * gk;
                v[k] = b2 * v[k] + (1.f - b2) * gk * gk;
                w[k] -= step_size * m[k] / (sqrtf(v[k]) * inv_bc2_sqrt + eps);
            }
        }
    }
}

int hvn_launch_adam(float *w, const float *g, float *m, float *v, long n, float b1, float b2, float eps, float step_size,
                    float inv_bc2_sqrt, hipStream_t stream)
{
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(hvn_adam, dim3((unsigned)blocks), dim3(256), 0, stream, w, g, m, v, n, b1, b2, eps, step_size, inv_bc2_sqrt);
    return launch_ok();
}
