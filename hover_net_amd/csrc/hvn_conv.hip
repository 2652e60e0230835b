// hvn_conv.hip -- fused implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
// Stands behind every nn.Conv2d (+ the BatchNorm/ReLU/residual/concat/crop ops around
// it) of /root/reference/models/hovernet/net_desc.py:101-145 except conv0 and the 1x1
// logit heads (hvn_net_ops.hip).
//
// GEMM view:  D[m][co] = sum_k A[m][k] * Wt[co][k],   m = (n, oy, ox) output pixel,
//             k = (tap, ci) with ci fastest.  Activations are channels-last, so one
//             k-chunk of 32 is 128 contiguous bytes of A; weights are pre-packed
//             [cout_pad][taps][cin] so the same holds for B.
// Tile:       128 pixels x {128,64,32} output channels per 256-thread workgroup,
//             BK = 32, double-buffered LDS.  The global loads of k-step t+1 are issued
//             RAW before the MFMAs of step t and only touched (prologue BN+ReLU, zero
//             padding select, ds_write) after them, so HBM/L2 latency hides under
//             >=1024 matrix-pipe cycles per wave.
// MFMA:       v_mfma_f32_32x32x2_f32, exact fp32 (== an fmaf chain).  Lane l feeds
//             A[i=l&31][k=l>>5] / B[k=l>>5][j=l&31]; the k-ORDER inside a chunk of 8
//             is permuted (lanes <32 take k 0..3, lanes >=32 take k 4..7) so that each
//             lane reads its 4 operands with ONE ds_read_b128 -- a sum is a sum.
// LDS:        rows padded to 36 floats: the 16-lane groups of ds_read_b128 and the
//             8-lane groups of ds_write_b128 then hit disjoint banks (MI355X_MICROARCH
//             section LDS).
// Epilogue:   accumulators are transposed through LDS (the k-loop is done with it) so
//             that every global access is 16 B per lane and 512 B contiguous per row:
//             bias / ReLU / residual add / block-closing BN-ReLU are applied on float4s.
// Grid:       1-D, XCD-aware: workgroups that share an A tile (same pixels, different
//             cout tile) get the same blockIdx % 8, i.e. the same XCD L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: HIP's float4 struct copies lower to memcpy -> scratch

#define BK 32
#define LDS_LD 36  // padded row length in floats

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 2) void hvn_conv_igemm_f32(const ConvArgs p)
{
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;  // staging passes (32 rows of 8 float4 per pass)
    constexpr int EP_LD = BN + 4;              // epilogue tile row length (floats)
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");
    static_assert(BM * EP_LD <= 2 * (BM + BN) * LDS_LD, "epilogue tile must fit in the staging buffers");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                       // [2][BM][LDS_LD]
    float *Bs = smem + 2 * BM * LDS_LD;     // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware tile mapping (see header)
    const int NT = p.n_tiles;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int n_tile = seq % NT;
    const int m_tile = (seq / NT) * 8 + xcd;
    if (m_tile >= (int)p.m_tiles) return;
    const unsigned m0 = (unsigned)m_tile * BM;
    const int n0 = n_tile * BN;
    const unsigned M = (unsigned)p.M;

    // ---- per-thread staging coordinates --------------------------------------------
    const int srow = tid >> 3;      // 0..31
    const int scol = (tid & 7) * 4; // float offset inside the 32-wide k chunk
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    long a_off[PA];
    int a_iy[PA], a_ix[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const unsigned m = m0 + srow + 32 * j;
        if (m < M) {
            const unsigned n = m / HoWo;
            const unsigned rem = m - n * HoWo;
            const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
            a_iy[j] = (int)oy * p.stride - p.pad_t;
            a_ix[j] = (int)ox * p.stride - p.pad_l;
            a_off[j] = (long)n * p.xsn + (long)a_iy[j] * p.xsy + (long)a_ix[j] * p.xsx + scol;
        } else {
            a_iy[j] = -(1 << 28);
            a_ix[j] = -(1 << 28);
            a_off[j] = 0;
        }
    }
    const long Ktot = (long)p.KH * p.KW * p.Cin;
    const float *wrow0 = p.w + (long)(n0 + srow) * Ktot + scol;
    const long wstep = 32 * Ktot;  // 32 weight rows further per staging pass

    const int kchunks = p.Cin / BK;
    const int KT = p.KH * p.KW * kchunks;
    const bool has_pre = p.pre_s != nullptr;
    // branch-free prologue: scale 1 / shift 0 / clamp -inf when there is none
    const float pre_lo = has_pre ? 0.f : -__builtin_inff();
    const float *pre_s = has_pre ? p.pre_s : p.w;  // any valid address; value unused when !has_pre
    const float *pre_b = has_pre ? p.pre_b : p.w;

    f32x4 ra[PA], rb[PB], rps, rpb;
    unsigned okmask = 0;
    int ld_r = 0, ld_s = 0, ld_c = 0;  // tap row / col / channel-chunk of the NEXT load

    // issue the raw loads of one k-step (nothing here waits on memory)
    auto load_global = [&](int kt) {
        const long tap_off = (long)ld_r * p.xsy + (long)ld_s * p.xsx + ld_c * BK;
        rps = *(const f32x4 *)(pre_s + (has_pre ? ld_c * BK + scol : 0));
        rpb = *(const f32x4 *)(pre_b + (has_pre ? ld_c * BK + scol : 0));
        okmask = 0;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const bool ok = (unsigned)(a_iy[j] + ld_r) < (unsigned)p.H && (unsigned)(a_ix[j] + ld_s) < (unsigned)p.W;
            const float *src = ok ? (p.x + a_off[j] + tap_off) : p.x;  // out-of-image taps read a safe address
            ra[j] = *(const f32x4 *)src;
            okmask |= ok ? (1u << j) : 0u;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) rb[j] = *(const f32x4 *)(wrow0 + j * wstep + (long)kt * BK);
        // advance (ci fastest, then tap column, then tap row)
        if (++ld_c == kchunks) {
            ld_c = 0;
            if (++ld_s == p.KW) {
                ld_s = 0;
                ++ld_r;
            }
        }
    };
    // first touch of the loaded registers: pre-activation BN+ReLU, zero padding, park in LDS
    auto store_lds = [&](int buf) {
        float *a = As + buf * BM * LDS_LD;
        float *b = Bs + buf * BN * LDS_LD;
        f32x4 ps = rps, pb = rpb;
        if (!has_pre) {
            ps = (f32x4){1.f, 1.f, 1.f, 1.f};
            pb = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const bool ok = (okmask >> j) & 1u;
            f32x4 v = ra[j];
            v.x = fmaxf(fmaf(v.x, ps.x, pb.x), pre_lo);
            v.y = fmaxf(fmaf(v.y, ps.y, pb.y), pre_lo);
            v.z = fmaxf(fmaf(v.z, ps.z, pb.z), pre_lo);
            v.w = fmaxf(fmaf(v.w, ps.w, pb.w), pre_lo);
            v.x = ok ? v.x : 0.f;
            v.y = ok ? v.y : 0.f;
            v.z = ok ? v.z : 0.f;
            v.w = ok ? v.w : 0.f;
            *(f32x4 *)(a + (srow + 32 * j) * LDS_LD + scol) = v;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) *(f32x4 *)(b + (srow + 32 * j) * LDS_LD + scol) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int cur) {
        const float *a = As + cur * BM * LDS_LD + (wm * WM + l31) * LDS_LD + 4 * lh;
        const float *b = Bs + cur * BN * LDS_LD + (wn * WN + l31) * LDS_LD + 4 * lh;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const f32x4 *)(a + i * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const f32x4 *)(b + j * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };

    load_global(0);
    store_lds(0);
    __syncthreads();
    // steady state: issue the global loads of step kt+1, run the MFMAs of step kt out of
    // LDS buffer kt&1, then park the loaded registers in the other buffer; one barrier per step
    for (int kt = 0; kt < KT - 1; ++kt) {
        load_global(kt + 1);
        compute(kt & 1);
        store_lds((kt + 1) & 1);
        __syncthreads();
    }
    compute((KT - 1) & 1);
    __syncthreads();

    // ---- epilogue ------------------------------------------------------------------
    // 1. accumulators -> LDS tile [BM][EP_LD] (32 consecutive floats per half-wave: conflict-free)
    float *ep = smem;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                ep[row * EP_LD + wn * WN + j * 32 + l31] = acc[i][j][r];
            }
    __syncthreads();
    // 2. each thread owns one float4 column chunk and walks rows: 16 B per lane, BN*4 B contiguous per row
    constexpr int CH = BN / 4;            // float4 chunks per row
    constexpr int RPP = 256 / CH;         // rows per pass
    const int ecol = (tid % CH) * 4;
    const int erow0 = tid / CH;
    const int co = n0 + ecol;
    const bool cok = co < p.Cout;         // Cout is a multiple of 4 (validated on the host)
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, qs = {1.f, 1.f, 1.f, 1.f}, qb = bias;
    const bool has_res = p.res != nullptr, has_post = p.post_s != nullptr;
    if (cok) {
        if (p.bias) bias = *(const f32x4 *)(p.bias + co);
        if (has_post) {
            qs = *(const f32x4 *)(p.post_s + co);
            qb = *(const f32x4 *)(p.post_b + co);
        }
    }
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    const float post_lo = has_post ? 0.f : -__builtin_inff();
    constexpr int NIT = BM / RPP;         // rows per thread: 16 / 8 / 4
    const float *resp = has_res ? p.res : p.w;  // valid address either way; value unused without a residual
    for (int it0 = 0; it0 < NIT; it0 += 4) {
        long yoff[4], roff[4];
        bool ok[4];
        f32x4 rv[4];
        // decode 4 rows, then issue their residual loads together, then do the math / stores
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned m = m0 + erow0 + (it0 + u) * RPP;
            ok[u] = m < M && cok;
            const unsigned mm = ok[u] ? m : 0u;
            const unsigned n = mm / HoWo;
            const unsigned rem = mm - n * HoWo;
            const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
            yoff[u] = (long)n * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + co;
            roff[u] = (ok[u] && has_res) ? (long)n * p.rsn + (long)oy * p.rsy + (long)ox * p.rsx + co : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) rv[u] = *(const f32x4 *)(resp + roff[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = erow0 + (it0 + u) * RPP;
            f32x4 v = *(const f32x4 *)(ep + rr * EP_LD + ecol);
            v.x = fmaxf(v.x + bias.x, relu_lo);
            v.y = fmaxf(v.y + bias.y, relu_lo);
            v.z = fmaxf(v.z + bias.z, relu_lo);
            v.w = fmaxf(v.w + bias.w, relu_lo);
            if (has_res) v += rv[u];
            v.x = fmaxf(fmaf(v.x, qs.x, qb.x), post_lo);
            v.y = fmaxf(fmaf(v.y, qs.y, qb.y), post_lo);
            v.z = fmaxf(fmaf(v.z, qs.z, qb.z), post_lo);
            v.w = fmaxf(fmaf(v.w, qs.w, qb.w), post_lo);
            if (ok[u]) *(f32x4 *)(p.y + yoff[u]) = v;
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_conv(const ConvArgs &a, hipStream_t stream)
{
    ConvArgs p = a;
    p.m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = (p.Cout + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    static bool attr_done = false;
    auto kern = hvn_conv_igemm_f32<BM, BN, WAVES_M, WAVES_N>;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_done = true;
    }
    const long groups = (p.m_tiles + 7) / 8;
    const long grid = groups * 8 * p.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hvn_launch_conv(const ConvArgs &a, int tile_n, hipStream_t stream)
{
    if (a.Cin % BK != 0 || a.Cin <= 0 || a.Cout % 4 != 0) return -1;
    if (a.M <= 0 || a.M >= (1L << 31) - 256) return -1;  // 32-bit pixel index arithmetic in the kernel
    switch (tile_n) {
    case 128: return launch_conv<128, 128, 2, 2>(a, stream);
    case 64: return launch_conv<128, 64, 4, 1>(a, stream);
    case 32: return launch_conv<128, 32, 4, 1>(a, stream);
    default: return -1;
    }
}
