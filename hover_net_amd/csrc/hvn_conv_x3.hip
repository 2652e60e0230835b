// hvn_conv_x3.hip -- the implicit-GEMM convolution of hvn_conv.hip with fp32 operands and fp32 accumulation, its PRODUCTS formed on
// the gfx950 bf16 matrix pipe from exact three-way bf16 splits of the fp32 operands ("bf16x3").
//
// Why: v_mfma_f32_32x32x2_f32 retires 2 reduction steps in 64 cycles, v_mfma_f32_32x32x16_bf16 16 steps in 32 -- the bf16 pipe
// is 16x the fp32 pipe (MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 matrix peak, ~2.5 PFLOP/s bf16).  An fp32 number is the exact sum
// of three bf16 numbers (8 + 8 + 8 significand bits: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), every difference exact in
// fp32), and a bf16 x bf16 product is exact in fp32, so
//     a * b = sum over the nine (plane of a, plane of b) pairs of exact partial products,
// accumulated in the MFMA's fp32 accumulator: the same dot product as the fp32 MFMA computes, in another summation order --
// at 9/16 of its matrix time (NTERMS = 9).  NTERMS = 6 drops the three pairs whose weight is <= 2^-24 of the product (m*l, l*m, l*l):
// 6/16 of the matrix time, per-product truncation of the order of one fp32 rounding.  Activations, weights (as their three planes),
// accumulators and outputs stay fp32 everywhere else: prologue BN+ReLU, bias / ReLU / residual / block BN-ReLU epilogue, views and
// padding are those of hvn_conv_igemm_f32 (reference geometry: /root/reference/models/hovernet/net_utils.py:155-266, net_desc.py:76-99).
//
// Layout: a k-step is 32 reduction elements (one tap of one 32-channel slab, 128 B of A per pixel, like the fp32 kernel).  The packed
// weights are [cout_pad][k-step][plane 3][32] bf16 (192 B per row and k-step, split on the host: engine.split_bf16x3).  LDS rows hold
// [plane 3][32] bf16 + 16 B of padding (pitch 208 B = 52 banks = 4 x 13: any 16 rows with distinct r mod 16 -- every lane group of
// ds_read_b128 -- cover all 64 banks).  One LDS buffer + one register stage: barrier, split-and-store the staged step, barrier,
// issue the next step's global loads, 72 (54) MFMAs per wave out of LDS; two or three workgroups per CU cover each other's stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define XK 32        // reduction elements per k-step
#define XPITCH 104   // LDS row pitch in bf16 elements (3 planes x 32 + 8)

// x = h + m + l exactly (RNE conversions; x - h and x - h - m are exact in fp32) for 2^-110 <= |x| < 3.38e38 and for 0: below, the low
// planes underflow bf16's denormal grid (absolute error < 2^-133); within 0.3 % of FLT_MAX h rounds to infinity -- a value no fp32
// accumulation of this network survives either (tests/test_x3_arithmetic.py pins both limits)
__device__ __forceinline__ void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l)
{
    h = (__bf16)x;
    const float r = x - (float)h;
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);
}
__device__ __forceinline__ uint32_t pack2(__bf16 a, __bf16 b)
{
    bf16x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, v);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PADDED, bool HAS_PRE, bool HAS_X2, int NTERMS>
__global__ __launch_bounds__(256, 2) void hvn_conv_igemm_x3(ConvArgs p)
{
    const uint16_t *pw = (const uint16_t *)p.w;
    if (p.nbatch > 1) {  // batched launch: one of nbatch independent problems per blockIdx.y (Winograd transform positions)
        p.x += (long)blockIdx.y * p.xb;
        pw += (long)blockIdx.y * p.wb;     // wb counts bf16 elements of the 3-plane packing
        p.y += (long)blockIdx.y * p.yb;
    }
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM / 32;                // A staging passes (32 rows of 8 float4 per pass)
    constexpr int NB = BN * 12 / 256;          // B staging: 12 16-byte pieces per row and k-step
    constexpr int EP_LD = BN + 4;              // epilogue tile row length (floats)
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");
    static_assert(NTERMS == 9 || NTERMS == 6, "nine exact partial products, or the six that carry > 2^-24 of the product");
    extern __shared__ __attribute__((aligned(16))) uint16_t xs[];
    uint16_t *As = xs;                  // [BM][XPITCH]
    uint16_t *Bs = xs + BM * XPITCH;    // [BN][XPITCH]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware tile mapping (hvn_conv.hip): the cout tiles of one pixel tile share blockIdx % 8, i.e. an XCD's L2
    const int NT = p.n_tiles;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int n_tile = seq % NT;
    // (multi-tap launches: neighbouring pixel tiles share input rows, so XCD k takes the k-th CONTIGUOUS eighth of them and finds the
    //  halo in its own L2; in round-robin order every XCD fetched it from HBM for itself -- measured 2.1 - 3.3 x the compulsory reads)
    const int m_tile = hvn_m_tile(xcd, seq / NT, (int)(gridDim.x / (8u * (unsigned)NT)), p.KH * p.KW > 1);
    if (m_tile >= (int)p.m_tiles) return;
    const unsigned m0 = (unsigned)m_tile * BM;
    const int n0 = n_tile * BN;
    const unsigned M = (unsigned)p.M;

    // ---- A staging coordinates: wave-uniform base moving with the k-step (SGPR) + loop-invariant per-thread byte offset; a
    //      voffset beyond num_records returns zeros (padding taps, rows past the batch) --------------------------------------------
    const int srow = tid >> 3;
    const int scol = (tid & 7) * 4;
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const unsigned n_blk = m0 / HoWo;
    const long padoff = (long)p.pad_t * p.xsy + (long)p.pad_l * p.xsx;
    constexpr unsigned OOB = 0x80000000u;
    unsigned a_voff[PA];
    int a_iy[PA], a_ix[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const unsigned m = m0 + srow + 32 * j;
        const bool ok = m < M;
        const unsigned mm = ok ? m : m0;
        const unsigned n = mm / HoWo;
        const unsigned rem = mm - n * HoWo;
        const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
        a_iy[j] = ok ? (int)oy * p.stride - p.pad_t : -(1 << 28);
        a_ix[j] = ok ? (int)ox * p.stride - p.pad_l : -(1 << 28);
        a_voff[j] = ok ? (unsigned)(((long)(n - n_blk) * p.xsn + (long)(oy * p.stride) * p.xsy + (long)(ox * p.stride) * p.xsx + scol) * 4) : OOB;
    }
    const float *xblk = p.x + (long)n_blk * p.xsn - padoff;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)xblk, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)pw, 0, 0x7fffffff, 0x00020000);
    unsigned a2_voff[PA];
    const float *x2blk = HAS_X2 ? p.x2 + (long)n_blk * p.x2sn : p.x;
    const __amdgpu_buffer_rsrc_t rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc((void *)x2blk, 0, 0x7fffffff, 0x00020000);
    if constexpr (HAS_X2) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const unsigned m = m0 + srow + 32 * j;
            const bool ok = m < M;
            const unsigned mm = ok ? m : m0;
            const unsigned n = mm / HoWo;
            const unsigned rem = mm - n * HoWo;
            const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
            a2_voff[j] = ok ? (unsigned)(((long)(n - n_blk) * p.x2sn + (long)(oy * p.stride2) * p.x2sy + (long)(ox * p.stride2) * p.x2sx + scol) * 4) : OOB;
        }
    }
    const int kchunks = p.Cin / XK;
    const int KT1 = p.KH * p.KW * kchunks;
    const int KT = KT1 + (HAS_X2 ? p.Cin2 / XK : 0);
    // ---- B staging: 16-byte piece c = tid + 256 j of the [BN][12] pieces of one k-step --------------------------------------------
    unsigned w_voff[NB], b_lds[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int c = tid + 256 * j;
        const int row = c / 12, piece = c - row * 12;
        w_voff[j] = (unsigned)(((long)(n0 + row) * KT * 192) + piece * 16);
        b_lds[j] = (unsigned)(row * (XPITCH * 2) + piece * 16);
    }
    const bool has_pre = HAS_PRE && p.pre_s != nullptr;

    f32x4 ra[PA], rps, rpb;
    u32x4 rb[NB];
    int ld_r = 0, ld_s = 0, ld_c = 0;  // tap row / col / channel slab of the NEXT load
    auto load_global = [&](int kt) {
        int a_soff = (int)(((long)ld_r * p.xsy + (long)ld_s * p.xsx + (long)ld_c * XK) * 4);
        const int w_soff = kt * 192;
        const bool second = HAS_X2 && kt >= KT1;
        if constexpr (HAS_X2) a_soff = second ? (kt - KT1) * (XK * 4) : a_soff;
        if constexpr (HAS_PRE) {
            if (has_pre) {
                rps = *(const f32x4 *)(p.pre_s + ld_c * XK + scol);
                rpb = *(const f32x4 *)(p.pre_b + ld_c * XK + scol);
            }
        }
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            unsigned vo = a_voff[j];
            if constexpr (PADDED) {
                const bool ok = (unsigned)(a_iy[j] + ld_r) < (unsigned)p.H && (unsigned)(a_ix[j] + ld_s) < (unsigned)p.W;
                vo = ok ? vo : OOB;
            }
            if constexpr (HAS_X2) {
                vo = second ? a2_voff[j] : vo;
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? rsrc_a2 : rsrc_a, vo, a_soff, 0));
            } else
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, a_soff, 0));
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[j], w_soff, 0);
        if (++ld_s == p.KW) {
            ld_s = 0;
            if (++ld_r == p.KH) {
                ld_r = 0;
                ++ld_c;
            }
        }
    };
    // first touch of the staged step: prologue BN+ReLU (fp32), exact three-way split, park the planes in LDS
    auto store_lds = [&]() {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            f32x4 v = ra[j];
            if constexpr (HAS_PRE) {
                if (has_pre) {
                    v.x = fmaxf(fmaf(v.x, rps.x, rpb.x), 0.f);
                    v.y = fmaxf(fmaf(v.y, rps.y, rpb.y), 0.f);
                    v.z = fmaxf(fmaf(v.z, rps.z, rpb.z), 0.f);
                    v.w = fmaxf(fmaf(v.w, rps.w, rpb.w), 0.f);
                }
            }
            __bf16 h[4], m[4], l[4];
            split3(v.x, h[0], m[0], l[0]);
            split3(v.y, h[1], m[1], l[1]);
            split3(v.z, h[2], m[2], l[2]);
            split3(v.w, h[3], m[3], l[3]);
            uint16_t *dst = As + (srow + 32 * j) * XPITCH + scol;
            *(u32x2 *)(dst) = (u32x2){pack2(h[0], h[1]), pack2(h[2], h[3])};
            *(u32x2 *)(dst + 32) = (u32x2){pack2(m[0], m[1]), pack2(m[2], m[3])};
            *(u32x2 *)(dst + 64) = (u32x2){pack2(l[0], l[1]), pack2(l[2], l[3])};
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) *(u32x4 *)((unsigned char *)Bs + b_lds[j]) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // lane (l31, lh) feeds row l31, k = 16 q + 8 lh .. + 7 of every plane: one ds_read_b128 per (tile, plane, q)
    auto compute = [&]() {
        const uint16_t *a = As + (wm * WM + l31) * XPITCH + 8 * lh;
        const uint16_t *b = Bs + (wn * WN + l31) * XPITCH + 8 * lh;
#pragma unroll
        for (int q = 0; q < XK / 16; ++q) {
            bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fa[i][pl] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(a + i * 32 * XPITCH + pl * 32 + q * 16));
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fb[j][pl] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(b + j * 32 * XPITCH + pl * 32 + q * 16));
            // smallest partial products first; (plane of a, plane of b) with 0 = high, 2 = low
#pragma unroll
            for (int s = 4; s >= 0; --s) {
                if (NTERMS == 6 && s > 2) continue;
#pragma unroll
                for (int pa = 2; pa >= 0; --pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb > 2) continue;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa], fb[j][pb], acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    load_global(0);
    for (int kt = 0; kt < KT; ++kt) {
        __syncthreads();             // every wave is done reading the previous step out of LDS
        store_lds();
        __syncthreads();
        if (kt + 1 < KT) load_global(kt + 1);   // lands under this step's MFMAs
        compute();
    }
    __syncthreads();

    // ---- epilogue (fp32, as hvn_conv_igemm_f32): accumulators -> LDS tile -> bias / ReLU / + residual / block BN-ReLU -> 16-byte stores
    float *ep = (float *)xs;
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
    constexpr int CH = BN / 4;            // float4 chunks per row
    constexpr int RPP = 256 / CH;         // rows per pass
    constexpr int NIT = BM / RPP;
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
    // addresses: 32-bit byte offsets from the sample of the tile's first row, stepped row to row, through buffer descriptors (hvn_conv_x3g.hip:
    // the 64-bit products per load / store were ~600 VALU per thread and tile); out-of-range offset = zeros loaded, store dropped
    constexpr unsigned EOOB = 0x80000000u;
    unsigned e_oy, e_ox, y_off, r_off;
    const unsigned e_nblk = m0 / HoWo;
    {
        const unsigned m = m0 + erow0;
        const unsigned e_n = m / HoWo;
        const unsigned rem = m - e_n * HoWo;
        e_oy = rem / (unsigned)p.Wo;
        e_ox = rem - e_oy * (unsigned)p.Wo;
        y_off = (unsigned)(((long)(e_n - e_nblk) * p.ysn + (long)e_oy * p.ysy + (long)e_ox * p.ysx + co) * 4);
        r_off = (unsigned)(((long)(e_n - e_nblk) * p.rsn + (long)e_oy * p.rsy + (long)e_ox * p.rsx + co) * 4);
    }
    const unsigned y_step = (unsigned)(RPP * p.ysx * 4), y_row = (unsigned)((p.ysy - (long)p.Wo * p.ysx) * 4), y_smp = (unsigned)((p.ysn - (long)p.Ho * p.ysy) * 4);
    const unsigned r_step = (unsigned)(RPP * p.rsx * 4), r_row = (unsigned)((p.rsy - (long)p.Wo * p.rsx) * 4), r_smp = (unsigned)((p.rsn - (long)p.Ho * p.rsy) * 4);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void *)(p.y + (long)e_nblk * p.ysn), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r =
        __builtin_amdgcn_make_buffer_rsrc((void *)(has_res ? p.res + (long)e_nblk * p.rsn : p.y + (long)e_nblk * p.ysn), 0, 0x7fffffff, 0x00020000);
    // all residual loads of the tile, then every value finished in registers, then the stores back to back (one vmcnt for loads
    // and stores on gfx9: hvn_conv.hip has the measurements)
    f32x4 rall[NIT], vout[NIT];
    unsigned yoffs[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const unsigned m = m0 + erow0 + it * RPP;
        const bool ok = m < M && cok;
        rall[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has_res) rall[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, ok ? r_off : EOOB, 0, 0));
        yoffs[it] = ok ? y_off : EOOB;
        e_ox += RPP;
        y_off += y_step;
        r_off += r_step;
        while (e_ox >= (unsigned)p.Wo) {
            e_ox -= (unsigned)p.Wo;
            ++e_oy;
            y_off += y_row;
            r_off += r_row;
        }
        while (e_oy >= (unsigned)p.Ho) {
            e_oy -= (unsigned)p.Ho;
            y_off += y_smp;
            r_off += r_smp;
        }
    }
    // Round 6 (hvn_conv_x3g.hip has the argument): the arithmetic in the 8 forms {bias + ReLU | neither} x {residual | none} x {block BN-ReLU |
    // none}; what a launch does not have was computed as an identity before (max(acc + 0, -inf), + 0, max(fma(., 1, 0), -inf)): same bits, 5 VALU
    // per output element whatever the launch needed.
    auto finish = [&](auto hb_t, auto hr_t, auto hp_t) {
        constexpr bool HB = decltype(hb_t)::value, HR = decltype(hr_t)::value, HP = decltype(hp_t)::value;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f32x4 v = *(const f32x4 *)(ep + (erow0 + it * RPP) * EP_LD + ecol);
            if constexpr (HB) {
                v.x = fmaxf(v.x + bias.x, relu_lo);
                v.y = fmaxf(v.y + bias.y, relu_lo);
                v.z = fmaxf(v.z + bias.z, relu_lo);
                v.w = fmaxf(v.w + bias.w, relu_lo);
            }
            if constexpr (HR) v += rall[it];
            if constexpr (HP) {
                v.x = fmaxf(fmaf(v.x, qs.x, qb.x), post_lo);
                v.y = fmaxf(fmaf(v.y, qs.y, qb.y), post_lo);
                v.z = fmaxf(fmaf(v.z, qs.z, qb.z), post_lo);
                v.w = fmaxf(fmaf(v.w, qs.w, qb.w), post_lo);
            }
            vout[it] = v;
        }
    };
    {
        using T = std::true_type;
        using F = std::false_type;
        const bool hb = p.bias != nullptr || p.relu;
#if defined(HVN_X3G_FULL_EPI) && HVN_X3G_FULL_EPI
        finish(T{}, T{}, T{});      // A/B build (lib.VARIANTS["fullepi"]): every operation, absent operands as identities
#else
        if (hb) {
            if (has_res) {
                if (has_post) finish(T{}, T{}, T{}); else finish(T{}, T{}, F{});
            } else {
                if (has_post) finish(T{}, F{}, T{}); else finish(T{}, F{}, F{});
            }
        } else {
            if (has_res) {
                if (has_post) finish(F{}, T{}, T{}); else finish(F{}, T{}, F{});
            } else {
                if (has_post) finish(F{}, F{}, T{}); else finish(F{}, F{}, F{});
            }
        }
#endif
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(vout[it]), "+v"(yoffs[it]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vout[it]), rsrc_y, yoffs[it], 0, 0);
}

// Training: the fp32 weight packings change every step, so their bf16 planes are made on the device.  src = fp32 packings, any
// concatenation of [...][32]-float granules (hvn_pack_w's forward / data-gradient / Winograd forms); dst = the same granules as
// [3][32] bf16 (element offset of a packing in dst = 3 x its offset in src).  One thread = 4 floats of one granule.
__global__ __launch_bounds__(256) void hvn_split_x3(const float *src, uint16_t *dst, long granules)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long g = t >> 3;
    const int q = (int)(t & 7);
    if (g >= granules) return;
    const f32x4 v = *(const f32x4 *)(src + g * 32 + q * 4);
    __bf16 h[4], m[4], l[4];
    split3(v.x, h[0], m[0], l[0]);
    split3(v.y, h[1], m[1], l[1]);
    split3(v.z, h[2], m[2], l[2]);
    split3(v.w, h[3], m[3], l[3]);
    uint16_t *d = dst + g * 96 + q * 4;
    *(u32x2 *)(d) = (u32x2){pack2(h[0], h[1]), pack2(h[2], h[3])};
    *(u32x2 *)(d + 32) = (u32x2){pack2(m[0], m[1]), pack2(m[2], m[3])};
    *(u32x2 *)(d + 64) = (u32x2){pack2(l[0], l[1]), pack2(l[2], l[3])};
}

int hvn_launch_split_x3(const float *src, uint16_t *dst, long granules, hipStream_t stream)
{
    if (!src || !dst || granules <= 0 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return -1;
    const long blocks = (granules * 8 + 255) / 256;
    if (blocks > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(hvn_split_x3, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, granules);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PADDED, bool HAS_PRE, bool HAS_X2, int NTERMS>
static int launch_x3(const ConvArgs &a, hipStream_t stream)
{
    ConvArgs p = a;
    p.m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = (p.Cout + BN - 1) / BN;
    constexpr size_t stage_b = (size_t)(BM + BN) * XPITCH * 2, ep_b = (size_t)BM * (BN + 4) * 4;
    const size_t lds = stage_b > ep_b ? stage_b : ep_b;
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_conv_igemm_x3<BM, BN, WAVES_M, WAVES_N, PADDED, HAS_PRE, HAS_X2, NTERMS>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long groups = (p.m_tiles + 7) / 8;
    const long grid = groups * 8 * p.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, p.nbatch > 1 ? p.nbatch : 1), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int NTERMS>
static int dispatch_x3(const ConvArgs &a, int tile_n, bool padded, hipStream_t stream)
{
    if (a.x2) {
        if (tile_n == 128) return launch_x3<128, 128, 2, 2, false, false, true, NTERMS>(a, stream);
        if (tile_n == 64) return launch_x3<128, 64, 4, 1, false, false, true, NTERMS>(a, stream);
        return -1;
    }
    switch (tile_n) {
    case 128:
        if (a.pre_s) return launch_x3<128, 128, 2, 2, false, true, false, NTERMS>(a, stream);
        return padded ? launch_x3<128, 128, 2, 2, true, false, false, NTERMS>(a, stream) : launch_x3<128, 128, 2, 2, false, false, false, NTERMS>(a, stream);
    case 64:
        if (a.pre_s) return launch_x3<128, 64, 4, 1, false, true, false, NTERMS>(a, stream);
        return padded ? launch_x3<128, 64, 4, 1, true, false, false, NTERMS>(a, stream) : launch_x3<128, 64, 4, 1, false, false, false, NTERMS>(a, stream);
    default: return -1;
    }
}

// a.w: the 3-plane bf16 packing [cout_pad][k-step][3][32] of the fp32 weights; everything else as hvn_launch_conv.
// terms: 9 (every partial product: the fp32 dot product in another summation order) or 6.
int hvn_launch_conv_x3(const ConvArgs &a, int tile_n, int terms, hipStream_t stream)
{
    if (a.Cin % XK != 0 || a.Cin <= 0 || a.Cout % 4 != 0 || a.groups > 1) return -1;
    if (a.M <= 0 || a.M >= (1L << 31) - 256) return -1;
    // a 128-row tile reaches (HoWo + 126) / HoWo samples ahead of its first row's sample (hvn_launch_conv): 32-bit offsets below 2^31
    const long howo = (long)a.Ho * a.Wo;
    if (howo <= 0) return -1;
    const long ahead = (howo + 126) / howo;
    const long span = ahead * a.xsn + (long)(a.H + a.KH) * a.xsy + (long)(a.W + a.KW) * a.xsx;
    if (span < 0 || span * 4 >= (1L << 31)) return -1;
    if (a.x2 && (ahead * a.x2sn + (long)a.H * a.x2sy * a.stride2) * 4 >= (1L << 31)) return -1;
    // the epilogue's 32-bit offsets into y / res, from the sample of the tile's first row
    if ((ahead * a.ysn + (long)(a.Ho + 1) * a.ysy + (long)a.Wo * a.ysx) * 4 >= (1L << 31)) return -1;
    if (a.res && (ahead * a.rsn + (long)(a.Ho + 1) * a.rsy + (long)a.Wo * a.rsx) * 4 >= (1L << 31)) return -1;
    const long kt = (long)a.KH * a.KW * (a.Cin / XK) + (a.x2 ? a.Cin2 / XK : 0);
    if ((long)(a.Cout + 128) * kt * 192 >= (1L << 31)) return -1;
    const bool padded = a.pad_t > 0 || a.pad_l > 0 || (a.Ho - 1) * a.stride - a.pad_t + a.KH > a.H ||
                        (a.Wo - 1) * a.stride - a.pad_l + a.KW > a.W;
    if (padded && a.pre_s) return -1;
    if (a.x2 && (padded || a.Cin2 % XK || a.pre_s)) return -1;
    return terms == 6 ? dispatch_x3<6>(a, tile_n, padded, stream) : dispatch_x3<9>(a, tile_n, padded, stream);
}
