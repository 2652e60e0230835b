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
//             BK = 32, double-buffered LDS, register-staged global loads issued one
//             k-step ahead of the MFMAs that hide them (fp32 MFMA: 64 cycles per
//             32x32x2 => a k-step is >=1024 matrix-pipe cycles per wave).
// MFMA:       v_mfma_f32_32x32x2_f32, exact fp32 (== an fmaf chain).  Lane l feeds
//             A[i=l&31][k=l>>5] / B[k=l>>5][j=l&31]; the k-ORDER inside a chunk of 8
//             is permuted (lanes <32 take k 0..3, lanes >=32 take k 4..7) so that each
//             lane reads its 4 operands with ONE ds_read_b128 -- a sum is a sum.
// LDS:        rows padded to 36 floats: the 16-lane groups of ds_read_b128 and the
//             8-lane groups of ds_write_b128 then hit disjoint banks (MI355X_MICROARCH
//             section LDS).
// Grid:       1-D, XCD-aware: workgroups that share an A tile (same pixels, different
//             cout tile) get the same blockIdx % 8, i.e. the same XCD L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32
#define LDS_LD 36  // padded row length in floats

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void hvn_conv_igemm_f32(const ConvArgs p)
{
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;  // staging passes (32 rows of 8 float4 per pass)
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");
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
    const long m_tile = (long)(seq / NT) * 8 + xcd;
    if (m_tile >= p.m_tiles) return;
    const long m0 = m_tile * BM;
    const int n0 = n_tile * BN;

    // ---- per-thread staging coordinates --------------------------------------------
    const int srow = tid >> 3;      // 0..31
    const int scol = (tid & 7) * 4; // float offset inside the 32-wide k chunk
    const int HoWo = p.Ho * p.Wo;
    long a_off[PA];
    int a_iy[PA], a_ix[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        long m = m0 + srow + 32 * j;
        if (m < p.M) {
            int n = (int)(m / HoWo);
            int rem = (int)(m - (long)n * HoWo);
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy[j] = oy * p.stride - p.pad_t;
            a_ix[j] = ox * p.stride - p.pad_l;
            a_off[j] = (long)n * p.xsn + (long)a_iy[j] * p.xsy + (long)a_ix[j] * p.xsx + scol;
        } else {
            a_iy[j] = -(1 << 28);
            a_ix[j] = -(1 << 28);
            a_off[j] = 0;
        }
    }
    const long Ktot = (long)p.KH * p.KW * p.Cin;
    const float *wrow[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) wrow[j] = p.w + (long)(n0 + srow + 32 * j) * Ktot + scol;

    const int kchunks = p.Cin / BK;
    const int KT = p.KH * p.KW * kchunks;
    const bool has_pre = p.pre_s != nullptr;

    float4 ra[PA], rb[PB];
    int ld_r = 0, ld_s = 0, ld_c = 0;  // tap row / col / channel-chunk of the NEXT load

    // Branch-free staging loads: out-of-image taps read a safe address and are zeroed by a
    // select, the prologue is always applied (scale 1 / shift 0 / clamp -inf when absent).
    const float pre_lo = has_pre ? 0.f : -__builtin_inff();
    auto load_global = [&](int kt) {
        const long tap_off = (long)ld_r * p.xsy + (long)ld_s * p.xsx + ld_c * BK;
        float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_pre) {
            ps = *(const float4 *)(p.pre_s + ld_c * BK + scol);
            pb = *(const float4 *)(p.pre_b + ld_c * BK + scol);
        }
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const bool ok = (unsigned)(a_iy[j] + ld_r) < (unsigned)p.H && (unsigned)(a_ix[j] + ld_s) < (unsigned)p.W;
            const float *src = ok ? (p.x + a_off[j] + tap_off) : p.x;
            float4 v = *(const float4 *)src;
            v.x = fmaxf(fmaf(v.x, ps.x, pb.x), pre_lo);
            v.y = fmaxf(fmaf(v.y, ps.y, pb.y), pre_lo);
            v.z = fmaxf(fmaf(v.z, ps.z, pb.z), pre_lo);
            v.w = fmaxf(fmaf(v.w, ps.w, pb.w), pre_lo);
            ra[j].x = ok ? v.x : 0.f;
            ra[j].y = ok ? v.y : 0.f;
            ra[j].z = ok ? v.z : 0.f;
            ra[j].w = ok ? v.w : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) rb[j] = *(const float4 *)(wrow[j] + (long)kt * BK);
        // advance (ci fastest, then tap column, then tap row)
        if (++ld_c == kchunks) {
            ld_c = 0;
            if (++ld_s == p.KW) {
                ld_s = 0;
                ++ld_r;
            }
        }
    };
    auto store_lds = [&](int buf) {
        float *a = As + buf * BM * LDS_LD;
        float *b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int j = 0; j < PA; ++j) *(float4 *)(a + (srow + 32 * j) * LDS_LD + scol) = ra[j];
#pragma unroll
        for (int j = 0; j < PB; ++j) *(float4 *)(b + (srow + 32 * j) * LDS_LD + scol) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_global(0);
    store_lds(0);
    __syncthreads();

    auto compute = [&](int cur) {
        const float *a = As + cur * BM * LDS_LD + (wm * WM + l31) * LDS_LD + 4 * lh;
        const float *b = Bs + cur * BN * LDS_LD + (wn * WN + l31) * LDS_LD + 4 * lh;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const float4 *)(a + i * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const float4 *)(b + j * 32 * LDS_LD + q * 8);
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
    // per-row output / residual offsets, decoded once into LDS (the k-loop is done with it)
    long *yo = (long *)smem;        // [BM]
    long *ro = yo + BM;             // [BM]
    if (tid < BM) {
        long m = m0 + tid;
        long o = -1, r = 0;
        if (m < p.M) {
            int n = (int)(m / HoWo);
            int rem = (int)(m - (long)n * HoWo);
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            o = (long)n * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx;
            r = (long)n * p.rsn + (long)oy * p.rsy + (long)ox * p.rsx;
        }
        yo[tid] = o;
        ro[tid] = r;
    }
    __syncthreads();
    const bool has_res = p.res != nullptr, has_post = p.post_s != nullptr, has_bias = p.bias != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + wn * WN + j * 32 + l31;
        const bool cok = co < p.Cout;
        const float bias = (has_bias && cok) ? p.bias[co] : 0.f;
        const float qs = (has_post && cok) ? p.post_s[co] : 1.f;
        const float qb = (has_post && cok) ? p.post_b[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const long o = yo[row];
                if (o < 0 || !cok) continue;
                float v = acc[i][j][r] + bias;
                if (p.relu) v = fmaxf(v, 0.f);
                if (has_res) v += p.res[ro[row] + co];
                if (has_post) v = fmaxf(fmaf(v, qs, qb), 0.f);
                p.y[o + co] = v;
            }
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
    if (a.Cin % BK != 0 || a.Cin <= 0) return -1;
    switch (tile_n) {
    case 128: return launch_conv<128, 128, 2, 2>(a, stream);
    case 64: return launch_conv<128, 64, 4, 1>(a, stream);
    case 32: return launch_conv<128, 32, 4, 1>(a, stream);
    default: return -1;
    }
}
