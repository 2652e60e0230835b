// hvn_conv_bf16.hip -- the implicit-GEMM convolution of hvn_conv.hip with bf16 activations / weights and fp32
// accumulation on the gfx950 bf16 matrix cores (BASELINE cfg 3: "fast" mode, batch 64, bf16).
//
// Same GEMM view, tile (128 pixels x {128,64,32} output channels per 256-thread workgroup), staging geometry and
// fusions (prologue BN+ReLU on the input, epilogue bias / ReLU / residual / block-closing BN-ReLU, fused 1x1
// shortcut as a second reduction source) as the fp32 kernel; what changes:
//   * a k-step is 64 channels (still 128 B per row, so loads / LDS rows keep their byte layout), one
//     v_mfma_f32_32x32x16_bf16 consumes the 16 B a lane reads with one ds_read_b128 (8 bf16; the k-labelling inside
//     a step is arbitrary as long as A and B agree, which they do: both read the same byte offsets of their rows);
//   * 16 MFMAs x 32 cycles = 512 matrix cycles per k-step and wave instead of 4096, far less than the HBM / L2
//     latency, so the global loads run THREE k-steps ahead through a ring of four register stages;
//   * input channel counts that are a multiple of 32 but not of 64 (the dense units' 288, 352, ...): the upper half
//     of the last k-step is loaded as zeros (buffer range check) and the packed weights carry zeros there;
//   * grouped convs (dense-unit conv2) run as block-diagonal dense GEMMs (tiny layers, the bf16 pipe is 16x faster);
//   * the epilogue works in fp32 on the LDS-transposed accumulators and rounds to bf16 (RNE, v_cvt_pk_bf16_f32) on
//     the store; residuals are read as bf16.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define BKH 64      // reduction elements per k-step
#define LDH 72      // LDS row pitch in bf16 elements (144 B: conflict-free ds_read_b128 / ds_write_b128)

__device__ inline float bf_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ inline float bf_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ inline uint32_t pack_bf(float a, float b)
{
    bf16x2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, h);
}

// WIDE = true (round 4, default): the loop structure of hvn_conv_x3.hip -- a k-step is TWO 64-channel chunks side by side in one LDS
// buffer (row pitch 272 B = 68 banks = 4 x 17: conflict-free ds_read_b128), barrier, store the staged step, barrier, issue the next
// step's global loads, 32 MFMAs per wave between barriers instead of 16; one register stage (two workgroups per CU cover each other's
// store phases).  WIDE = false: round 2's loop (64-channel k-steps, double-buffered LDS, ring of four register stages); HVN_BF16_LOOP=0.
#define LDW 136     // WIDE: LDS row pitch in bf16 elements (2 x 64 + 8)
template <int BM, int BN, int WAVES_M, int WAVES_N, bool PADDED, bool HAS_PRE, bool HAS_X2, bool WIDE>
__global__ __launch_bounds__(256, 2) void hvn_conv_igemm_bf16(const ConvArgs p)
{
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;  // staging passes (32 rows of 8 x 16 B per pass)
    constexpr int EP_LD = BN + 4;              // epilogue tile row length (floats)
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");
    static_assert(WIDE || BM * EP_LD * 4 <= 2 * (BM + BN) * LDH * 2, "epilogue tile must fit in the staging buffers");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem16[];
    uint16_t *As = smem16;                                        // [2][BM][LDH]   | WIDE: [BM][LDW]
    uint16_t *Bs = smem16 + (WIDE ? BM * LDW : 2 * BM * LDH);     // [2][BN][LDH]   | WIDE: [BN][LDW]
    const uint16_t *px = (const uint16_t *)p.x;
    const uint16_t *pw = (const uint16_t *)p.w;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

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

    const int srow = tid >> 3;       // 0..31
    const int scol = (tid & 7) * 8;  // element offset inside the 64-wide k chunk
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
        a_voff[j] = ok ? (unsigned)(((long)(n - n_blk) * p.xsn + (long)(oy * p.stride) * p.xsy + (long)(ox * p.stride) * p.xsx + scol) * 2) : OOB;
    }
    const uint16_t *xblk = px + (long)n_blk * p.xsn - padoff;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)xblk, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)pw, 0, 0x7fffffff, 0x00020000);
    unsigned a2_voff[PA];
    const uint16_t *x2blk = HAS_X2 ? (const uint16_t *)p.x2 + (long)n_blk * p.x2sn : px;
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
            a2_voff[j] = ok ? (unsigned)(((long)(n - n_blk) * p.x2sn + (long)(oy * p.stride2) * p.x2sy + (long)(ox * p.stride2) * p.x2sx + scol) * 2) : OOB;
        }
    }
    const int kchunks = (p.Cin + BKH - 1) / BKH;
    const bool tail_half = (p.Cin % BKH) != 0 && scol >= 32;       // this thread's 8 channels lie past Cin in the last chunk
    const int KT1 = p.KH * p.KW * kchunks;
    const int KT = KT1 + (HAS_X2 ? p.Cin2 / BKH : 0);
    const long Ktot = (long)KT * BKH;
    unsigned w_voff[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) w_voff[j] = (unsigned)(((long)(n0 + srow + 32 * j) * Ktot + scol) * 2);
    const bool has_pre = HAS_PRE && p.pre_s != nullptr;

    struct Stage {
        u32x4 ra[PA], rb[PB];
        int ch;   // HAS_PRE: first input channel of this thread's 8 (the prologue constants are fetched at store time)
    };
    Stage st[4];
    int ld_r = 0, ld_s = 0, ld_c = 0;  // tap row / col / channel chunk of the NEXT load

    auto load_global = [&](Stage &s, int kt) {
        const bool past = kt >= KT;                                    // (uniform; WIDE only) the odd tail's second half: zeros for A and B
        int a_soff = (int)(((long)ld_r * p.xsy + (long)ld_s * p.xsx + ld_c * BKH) * 2);
        const int w_soff = past ? 0 : kt * (BKH * 2);
        const bool second = HAS_X2 && kt >= KT1;
        if constexpr (HAS_X2) a_soff = second ? (kt - KT1) * (BKH * 2) : a_soff;
        if (past) a_soff = 0;
        const bool zero_half = past || (tail_half && ld_c == kchunks - 1 && !second);
        if constexpr (HAS_PRE) s.ch = zero_half ? 0 : ld_c * BKH + scol;      // any valid index for the zero-weighted tail
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            unsigned vo = a_voff[j];
            if constexpr (PADDED) {
                const bool ok = (unsigned)(a_iy[j] + ld_r) < (unsigned)p.H && (unsigned)(a_ix[j] + ld_s) < (unsigned)p.W;
                vo = ok ? vo : OOB;
            }
            vo = zero_half ? OOB : vo;
            if constexpr (HAS_X2) {
                vo = second ? a2_voff[j] : vo;
                vo = past ? OOB : vo;
                s.ra[j] = __builtin_amdgcn_raw_buffer_load_b128(second ? rsrc_a2 : rsrc_a, vo, a_soff, 0);
            } else
                s.ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, a_soff, 0);
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) s.rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, past ? OOB : w_voff[j], w_soff, 0);
        if (past) return;
        if (++ld_s == p.KW) {
            ld_s = 0;
            if (++ld_r == p.KH) {
                ld_r = 0;
                ++ld_c;
            }
        }
    };
    auto store_lds = [&](Stage &s, int buf) {    // WIDE: buf = which half of the row (0 | 1)
        constexpr int LD = WIDE ? LDW : LDH;
        uint16_t *a = WIDE ? As + buf * BKH : As + buf * BM * LDH;
        uint16_t *b = WIDE ? Bs + buf * BKH : Bs + buf * BN * LDH;
        f32x4 ps0, ps1, pb0, pb1;
        if constexpr (HAS_PRE) {
            if (has_pre) {      // 32 B each, L1/L2 resident (one [Cin] array pair per layer)
                ps0 = *(const f32x4 *)(p.pre_s + s.ch);
                ps1 = *(const f32x4 *)(p.pre_s + s.ch + 4);
                pb0 = *(const f32x4 *)(p.pre_b + s.ch);
                pb1 = *(const f32x4 *)(p.pre_b + s.ch + 4);
            }
        }
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            u32x4 v = s.ra[j];
            if constexpr (HAS_PRE) {
                if (has_pre) {  // pre-activation BN + ReLU on the 8 bf16 channels of this thread, in fp32
                    const float s_[8] = {ps0.x, ps0.y, ps0.z, ps0.w, ps1.x, ps1.y, ps1.z, ps1.w};
                    const float b_[8] = {pb0.x, pb0.y, pb0.z, pb0.w, pb1.x, pb1.y, pb1.z, pb1.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = fmaxf(fmaf(bf_lo(v[e]), s_[2 * e], b_[2 * e]), 0.f);
                        const float hi = fmaxf(fmaf(bf_hi(v[e]), s_[2 * e + 1], b_[2 * e + 1]), 0.f);
                        v[e] = pack_bf(lo, hi);
                    }
                }
            }
            *(u32x4 *)(a + (srow + 32 * j) * LD + scol) = v;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) *(u32x4 *)(b + (srow + 32 * j) * LD + scol) = s.rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int cur) {
        constexpr int LD = WIDE ? LDW : LDH;
        const uint16_t *a = (WIDE ? As : As + cur * BM * LDH) + (wm * WM + l31) * LD + 8 * lh;
        const uint16_t *b = (WIDE ? Bs : Bs + cur * BN * LDH) + (wn * WN + l31) * LD + 8 * lh;
#pragma unroll
        for (int q = 0; q < (WIDE ? 2 : 1) * BKH / 16; ++q) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(a + i * 32 * LD + q * 16));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(b + j * 32 * LD + q * 16));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (WIDE) {
        // one LDS buffer, one register stage of two chunks: barrier, park the staged pair, barrier, issue the next pair's loads, compute
        load_global(st[0], 0);
        load_global(st[1], 1);                   // kt >= KT: zeros (odd tail)
        for (int kt = 0; kt < KT; kt += 2) {
            __syncthreads();
            store_lds(st[0], 0);
            store_lds(st[1], 1);
            __syncthreads();
            if (kt + 2 < KT) {
                load_global(st[0], kt + 2);
                load_global(st[1], kt + 3);
            }
            compute(0);
        }
        __syncthreads();
    } else {
    // ring of four register stages: the loads of k-step t+3 are issued while step t computes
    load_global(st[0], 0);
    if (KT > 1) load_global(st[1], 1);
    if (KT > 2) load_global(st[2], 2);
    store_lds(st[0], 0);
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = kt + u;
            if (t < KT) {
                if (t + 3 < KT) load_global(st[(u + 3) & 3], t + 3);
                compute(t & 1);
                if (t + 1 < KT) store_lds(st[(u + 1) & 3], (t + 1) & 1);
                __syncthreads();
            }
        }
    }
    }

    // ---- epilogue: accumulators -> fp32 LDS tile -> bias / ReLU / residual / post BN-ReLU -> bf16 ----
    float *ep = (float *)smem16;
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
    constexpr int CH = BN / 8;            // 8 output channels (16 B of bf16) per thread
    constexpr int RPP = 256 / CH;
    const int ecol = (tid % CH) * 8;
    const int erow0 = tid / CH;
    const int co = n0 + ecol;
    const bool cok = co < p.Cout;         // Cout is a multiple of 8 on this path (validated on the host)
    f32x4 bias[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, qs[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, qb[2] = {bias[0], bias[0]};
    const bool has_res = p.res != nullptr, has_post = p.post_s != nullptr;
    if (cok) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (p.bias) bias[h] = *(const f32x4 *)(p.bias + co + 4 * h);
            if (has_post) {
                qs[h] = *(const f32x4 *)(p.post_s + co + 4 * h);
                qb[h] = *(const f32x4 *)(p.post_b + co + 4 * h);
            }
        }
    }
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    const float post_lo = has_post ? 0.f : -__builtin_inff();
    uint16_t *py = (uint16_t *)p.y;
    const uint16_t *pres = (const uint16_t *)p.res;
    if (erow0 < BM) {
        // rows m0 + erow0 + it * RPP: the first located with divisions, the rest walked incrementally; addresses = 32-bit byte offsets from the
        // sample of the tile's first row through buffer descriptors (hvn_conv_x3g.hip: the 64-bit products per load / store cost several times
        // the arithmetic they served); out-of-range offset = zeros loaded, store dropped
        constexpr unsigned EOOB = 0x80000000u;
        unsigned oy, ox, y_off, r_off;
        const unsigned e_nblk = m0 / HoWo;
        {
            const unsigned m = m0 + erow0;
            const unsigned n = m / HoWo;
            const unsigned rem = m - n * HoWo;
            oy = rem / (unsigned)p.Wo;
            ox = rem - oy * (unsigned)p.Wo;
            y_off = (unsigned)(((long)(n - e_nblk) * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + co) * 2);
            r_off = (unsigned)(((long)(n - e_nblk) * p.rsn + (long)oy * p.rsy + (long)ox * p.rsx + co) * 2);
        }
        const unsigned y_step = (unsigned)(RPP * p.ysx * 2), y_row = (unsigned)((p.ysy - (long)p.Wo * p.ysx) * 2), y_smp = (unsigned)((p.ysn - (long)p.Ho * p.ysy) * 2);
        const unsigned r_step = (unsigned)(RPP * p.rsx * 2), r_row = (unsigned)((p.rsy - (long)p.Wo * p.rsx) * 2), r_smp = (unsigned)((p.rsn - (long)p.Ho * p.rsy) * 2);
        const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void *)(py + (long)e_nblk * p.ysn), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_r =
            __builtin_amdgcn_make_buffer_rsrc((void *)(has_res ? pres + (long)e_nblk * p.rsn : py + (long)e_nblk * p.ysn), 0, 0x7fffffff, 0x00020000);
        // all residual loads of the tile first, then the stores back to back: vmcnt retires loads and stores in order, so a
        // load issued after a store cannot be waited for without draining that store (see hvn_conv.hip)
        constexpr int NIT = (BM + RPP - 1) / RPP;
        u32x4 rall[NIT], vout[NIT];
        unsigned yoffs[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = erow0 + it * RPP;
            const bool ok = rr < BM && m0 + rr < M && cok;
            rall[it] = (u32x4){0u, 0u, 0u, 0u};
            if (has_res) rall[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, ok ? r_off : EOOB, 0, 0);
            yoffs[it] = ok ? y_off : EOOB;
            ox += RPP;
            y_off += y_step;
            r_off += r_step;
            while (ox >= (unsigned)p.Wo) {
                ox -= (unsigned)p.Wo;
                ++oy;
                y_off += y_row;
                r_off += r_row;
            }
            while (oy >= (unsigned)p.Ho) {
                oy -= (unsigned)p.Ho;
                y_off += y_smp;
                r_off += r_smp;
            }
        }
        // Round 6 (hvn_conv_x3g.hip has the argument): the arithmetic in the 8 forms {bias + ReLU | neither} x {residual | none} x {block
        // BN-ReLU | none}; what a launch does not have used to be computed as an identity (max(acc + 0, -inf), max(fma(., 1, 0), -inf)): same bits.
        auto finish = [&](auto hb_t, auto hr_t, auto hp_t) {
            constexpr bool HB = decltype(hb_t)::value, HR = decltype(hr_t)::value, HP = decltype(hp_t)::value;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int rr = erow0 + it * RPP;
                const u32x4 r4 = rall[it];
                u32x4 o;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 v = *(const f32x4 *)(ep + (rr < BM ? rr : 0) * EP_LD + ecol + 4 * h);
                    if constexpr (HB) {
                        v.x = fmaxf(v.x + bias[h].x, relu_lo);
                        v.y = fmaxf(v.y + bias[h].y, relu_lo);
                        v.z = fmaxf(v.z + bias[h].z, relu_lo);
                        v.w = fmaxf(v.w + bias[h].w, relu_lo);
                    }
                    if (HR && has_res) {
                        v.x += bf_lo(r4[2 * h]);
                        v.y += bf_hi(r4[2 * h]);
                        v.z += bf_lo(r4[2 * h + 1]);
                        v.w += bf_hi(r4[2 * h + 1]);
                    }
                    if constexpr (HP) {
                        v.x = fmaxf(fmaf(v.x, qs[h].x, qb[h].x), post_lo);
                        v.y = fmaxf(fmaf(v.y, qs[h].y, qb[h].y), post_lo);
                        v.z = fmaxf(fmaf(v.z, qs[h].z, qb[h].z), post_lo);
                        v.w = fmaxf(fmaf(v.w, qs[h].w, qb[h].w), post_lo);
                    }
                    o[2 * h] = pack_bf(v.x, v.y);
                    o[2 * h + 1] = pack_bf(v.z, v.w);
                }
                vout[it] = o;
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
        // ... and the stores leave back to back (the empty asm keeps LLVM from sinking the arithmetic into the store blocks, which
        // would put a vmcnt(0) -- a wait for the previous STORE -- between them; hvn_conv.hip has the measurements)
#pragma unroll
        for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(vout[it]), "+v"(yoffs[it]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) __builtin_amdgcn_raw_buffer_store_b128(vout[it], rsrc_y, yoffs[it], 0, 0);
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PADDED, bool HAS_PRE, bool HAS_X2, bool WIDE>
static int launch_bf16w(const ConvArgs &a, hipStream_t stream)
{
    ConvArgs p = a;
    p.m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = (p.Cout + BN - 1) / BN;
    constexpr size_t stage_b = WIDE ? (size_t)(BM + BN) * LDW * 2 : (size_t)2 * (BM + BN) * LDH * 2, ep_b = (size_t)BM * (BN + 4) * 4;
    const size_t lds = stage_b > ep_b ? stage_b : ep_b;
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_conv_igemm_bf16<BM, BN, WAVES_M, WAVES_N, PADDED, HAS_PRE, HAS_X2, WIDE>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long groups = (p.m_tiles + 7) / 8;
    const long grid = groups * 8 * p.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PADDED, bool HAS_PRE, bool HAS_X2>
static int launch_bf16(const ConvArgs &a, hipStream_t stream)
{
    static int wide = -1;
    if (wide < 0) {
        const char *e = getenv("HVN_BF16_LOOP");       // 0: round 2's loop (A/B runs); default: the wide-step loop
        wide = e ? atoi(e) : 1;
    }
    // a reduction of ONE 64-channel chunk (d0's conv3: K = 64) would pair it with a chunk of zeros: the narrow loop keeps it
    const long kt = (long)a.KH * a.KW * ((a.Cin + BKH - 1) / BKH) + (a.x2 ? a.Cin2 / BKH : 0);
    return (wide && kt >= 2) ? launch_bf16w<BM, BN, WAVES_M, WAVES_N, PADDED, HAS_PRE, HAS_X2, true>(a, stream)
                             : launch_bf16w<BM, BN, WAVES_M, WAVES_N, PADDED, HAS_PRE, HAS_X2, false>(a, stream);
}

int hvn_launch_conv_bf16(const ConvArgs &a, int tile_n, hipStream_t stream)
{
    if (a.Cin % 32 != 0 || a.Cin <= 0 || a.Cout % 8 != 0 || a.nbatch > 1) return -1;
    if ((((uintptr_t)a.y) & 15) || ((a.ysn | a.ysy | a.ysx) & 7) || (a.res && ((((uintptr_t)a.res) & 15) || ((a.rsn | a.rsy | a.rsx) & 7)))) return -1;   // 16-byte epilogue accesses
    if (a.M <= 0 || a.M >= (1L << 31) - 256) return -1;
    const long howo = (long)a.Ho * a.Wo;
    if (howo <= 0) return -1;
    const long ahead = (howo + 126) / howo;      // samples a 128-row tile reaches beyond its first row's (hvn_launch_conv)
    const long span = ahead * a.xsn + (long)(a.H + a.KH) * a.xsy + (long)(a.W + a.KW) * a.xsx;
    if (span < 0 || span * 2 >= (1L << 31)) return -1;
    // the epilogue's 32-bit offsets into y / res, from the sample of the tile's first row
    if ((ahead * a.ysn + (long)(a.Ho + 1) * a.ysy + (long)a.Wo * a.ysx) * 2 >= (1L << 31)) return -1;
    if (a.res && (ahead * a.rsn + (long)(a.Ho + 1) * a.rsy + (long)a.Wo * a.rsx) * 2 >= (1L << 31)) return -1;
    const long kt = (long)a.KH * a.KW * ((a.Cin + BKH - 1) / BKH) + (a.x2 ? a.Cin2 / BKH : 0);
    if ((long)(a.Cout + 128) * kt * BKH * 2 >= (1L << 31)) return -1;
    const bool padded = a.pad_t > 0 || a.pad_l > 0 || (a.Ho - 1) * a.stride - a.pad_t + a.KH > a.H ||
                        (a.Wo - 1) * a.stride - a.pad_l + a.KW > a.W;
    if (padded && a.pre_s) return -1;
    if (a.x2) {
        if (padded || a.Cin2 % BKH || a.Cin % BKH || a.pre_s) return -1;
        if (tile_n == 128) return launch_bf16<128, 128, 2, 2, false, false, true>(a, stream);
        if (tile_n == 64) return launch_bf16<128, 64, 4, 1, false, false, true>(a, stream);
        return -1;
    }
    switch (tile_n) {
    case 128:
        if (a.pre_s) return launch_bf16<128, 128, 2, 2, false, true, false>(a, stream);
        return padded ? launch_bf16<128, 128, 2, 2, true, false, false>(a, stream) : launch_bf16<128, 128, 2, 2, false, false, false>(a, stream);
    case 64:
        if (a.pre_s) return launch_bf16<128, 64, 4, 1, false, true, false>(a, stream);
        return padded ? launch_bf16<128, 64, 4, 1, true, false, false>(a, stream) : launch_bf16<128, 64, 4, 1, false, false, false>(a, stream);
    case 32:
        if (a.pre_s) return launch_bf16<128, 32, 4, 1, false, true, false>(a, stream);
        return padded ? launch_bf16<128, 32, 4, 1, true, false, false>(a, stream) : launch_bf16<128, 32, 4, 1, false, false, false>(a, stream);
    default: return -1;
    }
}
