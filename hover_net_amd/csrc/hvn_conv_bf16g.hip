// hvn_conv_bf16g.hip -- the bf16 implicit-GEMM convolution of hvn_conv_bf16.hip (BASELINE cfg 3: 'fast' mode, batch 64, bf16 activations
// and weights, fp32 accumulation) with BOTH operands staged by LDS-DMA (`buffer_load_dwordx4 ... lds`), as hvn_conv_x3g.hip does for the
// fp32 path: no staging registers, no ds_write pass, rings of three (activations) and three | two (weights) 64-channel stages with a
// counted s_waitcnt that leaves the youngest stages in flight across the ONE barrier of a k-step.  A k-step = 64 channels of one tap:
// 128 B per pixel row and per weight row, 16-byte piece p of row r at p ^ ((r >> 1) & 7) on the SOURCE side (LDS-DMA writes
// lane-linear) -- conflict-free ds_read_b128 for both operands without padding.  Zero padding, rows past the batch and the upper half of
// a 32-channel tail are buffer range checks (an offset beyond num_records loads zeros into LDS).
// Workgroups: 256 pixels x 128 channels, 512 threads (one per CU: 96 + 48 KB of rings), or 128 x 128, 256 threads (two per CU: 48 + 32 KB).
// Every output element sums its products in hvn_conv_igemm_bf16's order (k ascending in groups of 16): BIT-IDENTICAL to it
// (tests/test_gpu_bf16.py), so the engine picks per launch shape by time.  Launches with a prologue (pre-activation BN on the input) or
// fewer than 128 output channels stay on hvn_conv_bf16.hip.  Reference geometry: /root/reference/models/hovernet/net_desc.py:76-99.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define HK 64               // reduction elements (channels) per k-step
#define HBN 128             // output channels per workgroup
#define H_BSTAGE (HBN * 128)      // bytes of one weight stage: [128 rows][64 bf16]

typedef __attribute__((address_space(3))) void *lds_ptr_t;

// one LDS-DMA instruction (hvn_conv_x3g.hip:dma16; the builtin exists in the device pass only)
__device__ __forceinline__ void hdma16(__amdgpu_buffer_rsrc_t rsrc, lds_ptr_t dst, unsigned voff, int soff)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
#endif
}
__device__ __forceinline__ float hbf_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float hbf_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ __forceinline__ uint32_t hpack_bf(float a, float b)
{
    bf16x2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, h);
}

template <int BM, bool PADDED, bool HAS_X2>
__global__ __launch_bounds__(BM * 2, 2) void hvn_conv_igemm_bf16g(ConvArgs p)
{
    constexpr int NTHR = BM * 2;                 // 256 | 512
    constexpr int NW = NTHR / 64;
    constexpr int WAVES_N = 2;
    constexpr int NA = 3;                        // activation ring depth
    constexpr int NB = BM == 256 ? 3 : 2;        // weight ring depth
    constexpr int A_STAGE = BM * 128;
    constexpr int GA = A_STAGE / 1024 / NW;      // DMA instructions per wave and activation stage: 4
    constexpr int GB = H_BSTAGE / 1024 / NW;     // per weight stage: 2 | 4
    constexpr int B_OFF = NA * A_STAGE;
    constexpr int EP_LD = HBN + 4;
    static_assert(GA == 4, "the counted waits below are written for four activation DMAs per wave and stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char hs[];
    const uint16_t *px = (const uint16_t *)p.x;
    const uint16_t *pw = (const uint16_t *)p.w;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int n0 = n_tile * HBN;
    const unsigned M = (unsigned)p.M;

    // ---- activation staging: DMA instruction jj of this wave fills slots (wave GA + jj) 64 + lane; slot = row * 8 + physical piece, holding
    //      the row's logical piece phys ^ ((row >> 1) & 7) (8 bf16 channels) -------------------------------------------------------------
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const unsigned n_blk = m0 / HoWo;
    const long padoff = (long)p.pad_t * p.xsy + (long)p.pad_l * p.xsx;
    constexpr unsigned OOB = 0x80000000u;
    unsigned a_voff[GA], a2_voff[GA];
    int a_iy[GA], a_ix[GA];
    bool a_upper[GA];                  // this lane's piece is the upper 32 channels of its 64-channel chunk
#pragma unroll
    for (int jj = 0; jj < GA; ++jj) {
        const int row = (wave * GA + jj) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        a_upper[jj] = piece >= 4;
        const unsigned m = m0 + row;
        const bool ok = m < M;
        const unsigned mm = ok ? m : m0;
        const unsigned n = mm / HoWo;
        const unsigned rem = mm - n * HoWo;
        const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
        a_iy[jj] = ok ? (int)oy * p.stride - p.pad_t : -(1 << 28);
        a_ix[jj] = ok ? (int)ox * p.stride - p.pad_l : -(1 << 28);
        a_voff[jj] = ok ? (unsigned)(((long)(n - n_blk) * p.xsn + (long)(oy * p.stride) * p.xsy + (long)(ox * p.stride) * p.xsx + piece * 8) * 2) : OOB;
        a2_voff[jj] = OOB;
        if constexpr (HAS_X2)
            a2_voff[jj] = ok ? (unsigned)(((long)(n - n_blk) * p.x2sn + (long)(oy * p.stride2) * p.x2sy + (long)(ox * p.stride2) * p.x2sx + piece * 8) * 2) : OOB;
    }
    const uint16_t *xblk = px + (long)n_blk * p.xsn - padoff;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)xblk, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)pw, 0, 0x7fffffff, 0x00020000);
    const uint16_t *x2blk = HAS_X2 ? (const uint16_t *)p.x2 + (long)n_blk * p.x2sn : px;
    const __amdgpu_buffer_rsrc_t rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc((void *)x2blk, 0, 0x7fffffff, 0x00020000);
    const int kchunks = (p.Cin + HK - 1) / HK;
    const bool tail32 = (p.Cin % HK) != 0;                     // the last chunk of a tap holds 32 channels: its upper half is zeros
    const int KT1 = p.KH * p.KW * kchunks;
    const int KT = KT1 + (HAS_X2 ? p.Cin2 / HK : 0);
    // ---- weight staging: instruction t = wave GB + jj fills slots t 64 + lane; slot = row * 8 + physical piece ---------------------------
    unsigned w_voff[GB];
#pragma unroll
    for (int jj = 0; jj < GB; ++jj) {
        const int row = (wave * GB + jj) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        w_voff[jj] = (unsigned)((long)(n0 + row) * KT * 128 + piece * 16);
    }

    int ld_r = 0, ld_s = 0, ld_c = 0;  // tap row / col / channel chunk of the NEXT activation stage to issue
    auto issue_a = [&](int kt) {
        int a_soff = __builtin_amdgcn_readfirstlane((int)(((long)ld_r * p.xsy + (long)ld_s * p.xsx + (long)ld_c * HK) * 2));
        const bool second = HAS_X2 && kt >= KT1;
        if constexpr (HAS_X2) a_soff = second ? (kt - KT1) * (HK * 2) : a_soff;
        const bool zero_upper = tail32 && ld_c == kchunks - 1 && !second;       // uniform
        const int stage = kt % NA;
#pragma unroll
        for (int jj = 0; jj < GA; ++jj) {
            unsigned vo = a_voff[jj];
            if constexpr (PADDED) {
                const bool ok = (unsigned)(a_iy[jj] + ld_r) < (unsigned)p.H && (unsigned)(a_ix[jj] + ld_s) < (unsigned)p.W;
                vo = ok ? vo : OOB;
            }
            vo = (zero_upper && a_upper[jj]) ? OOB : vo;
            lds_ptr_t dst = (lds_ptr_t)(hs + stage * A_STAGE + (wave * GA + jj) * 1024);
            if constexpr (HAS_X2) {
                if (second)
                    hdma16(rsrc_a2, dst, a2_voff[jj], a_soff);
                else
                    hdma16(rsrc_a, dst, vo, a_soff);
            } else
                hdma16(rsrc_a, dst, vo, a_soff);
        }
        if (++ld_s == p.KW) {
            ld_s = 0;
            if (++ld_r == p.KH) {
                ld_r = 0;
                ++ld_c;
            }
        }
    };
    auto issue_b = [&](int kt) {
        const int w_soff = kt * 128;
        const int stage = kt % NB;
#pragma unroll
        for (int jj = 0; jj < GB; ++jj) {
            lds_ptr_t dst = (lds_ptr_t)(hs + B_OFF + stage * H_BSTAGE + (wave * GB + jj) * 1024);
            hdma16(rsrc_w, dst, w_voff[jj], w_soff);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: lane (l31, lh) feeds row l31 of a 32-row tile, k = 16 q + 8 lh .. + 7 = logical piece 2 q + lh
    const int key = (l31 >> 1) & 7;
    const unsigned a_row = (unsigned)((wm * 64 + l31) * 128);
    const unsigned b_row = (unsigned)((wn * 64 + l31) * 128);
    auto compute = [&](int kt) {
        const unsigned char *as = hs + (kt % NA) * A_STAGE + a_row;
        const unsigned char *bs = hs + B_OFF + (kt % NB) * H_BSTAGE + b_row;
#pragma unroll
        for (int q = 0; q < HK / 16; ++q) {
            bf16x8 fa[2], fb[2];
            const int off = ((2 * q + lh) ^ key) << 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(as + i * 32 * 128 + off));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(bs + j * 32 * 128 + off));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- pipeline: activations three stages deep; weights three (BM = 256) or two.  In every step the weights are issued BEFORE the
    //      activations, so the counted wait at the step's end leaves exactly the youngest stages in flight across the barrier ----------------
    issue_a(0);
    issue_b(0);
    if (KT > 1) {
        if constexpr (NB == 3) issue_b(1);
        issue_a(1);
        if constexpr (NB == 3)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GA + GB) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 2 < KT;
        if constexpr (NB == 3) {
            if (more) issue_b(kt + 2);
        } else {
            if (kt + 1 < KT) issue_b(kt + 1);
        }
        if (more) issue_a(kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(kt);
        __builtin_amdgcn_sched_barrier(0);
        // stage kt + 1 has landed (this wave's share), this wave's reads of stage kt have returned; then the barrier
        if (more) {
            if constexpr (NB == 3)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GA + GB) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue (hvn_conv_igemm_bf16's): accumulators -> fp32 LDS tile -> bias / ReLU / residual / post BN-ReLU -> bf16 ----
    float *ep = (float *)hs;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                ep[row * EP_LD + wn * 64 + j * 32 + l31] = acc[i][j][r];
            }
    __syncthreads();
    constexpr int CH = HBN / 8;           // 8 output channels (16 B of bf16) per thread
    constexpr int RPP = NTHR / CH;        // 16 | 32 rows per pass
    constexpr int NIT = BM / RPP;         // 8
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
    // addresses: 32-bit byte offsets from the sample of the tile's first row, stepped row to row, through buffer descriptors (hvn_conv_x3g.hip:
    // the 64-bit products per load / store cost several times the arithmetic they served); out-of-range offset = zeros loaded, store dropped
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
    // all residual loads of the tile first, every value finished in registers, then the stores back to back (one vmcnt for loads and stores)
    u32x4 rall[NIT];
    unsigned yoffs[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = erow0 + it * RPP;
        const bool ok = m0 + rr < M && cok;
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
    u32x4 vout[NIT];
    // Round 6 (hvn_conv_x3g.hip has the argument): the arithmetic in the 8 forms {bias + ReLU | neither} x {residual | none} x {block BN-ReLU |
    // none}; what a launch does not have used to be computed as an identity (max(acc + 0, -inf), max(fma(., 1, 0), -inf)): same bits, and on the
    // bf16 path -- 16 MFMAs per k-step -- the epilogue's VALU is a visible share of a short-K launch.
    auto finish = [&](auto hb_t, auto hr_t, auto hp_t) {
        constexpr bool HB = decltype(hb_t)::value, HR = decltype(hr_t)::value, HP = decltype(hp_t)::value;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = erow0 + it * RPP;
            const u32x4 r4 = rall[it];
            u32x4 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 v = *(const f32x4 *)(ep + rr * EP_LD + ecol + 4 * h);
                if constexpr (HB) {
                    v.x = fmaxf(v.x + bias[h].x, relu_lo);
                    v.y = fmaxf(v.y + bias[h].y, relu_lo);
                    v.z = fmaxf(v.z + bias[h].z, relu_lo);
                    v.w = fmaxf(v.w + bias[h].w, relu_lo);
                }
                if (HR && has_res) {
                    v.x += hbf_lo(r4[2 * h]);
                    v.y += hbf_hi(r4[2 * h]);
                    v.z += hbf_lo(r4[2 * h + 1]);
                    v.w += hbf_hi(r4[2 * h + 1]);
                }
                if constexpr (HP) {
                    v.x = fmaxf(fmaf(v.x, qs[h].x, qb[h].x), post_lo);
                    v.y = fmaxf(fmaf(v.y, qs[h].y, qb[h].y), post_lo);
                    v.z = fmaxf(fmaf(v.z, qs[h].z, qb[h].z), post_lo);
                    v.w = fmaxf(fmaf(v.w, qs[h].w, qb[h].w), post_lo);
                }
                o[2 * h] = hpack_bf(v.x, v.y);
                o[2 * h + 1] = hpack_bf(v.z, v.w);
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
#pragma unroll
    for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(vout[it]), "+v"(yoffs[it]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) __builtin_amdgcn_raw_buffer_store_b128(vout[it], rsrc_y, yoffs[it], 0, 0);
}

template <int BM, bool PADDED, bool HAS_X2>
static int launch_bf16g(const ConvArgs &a, hipStream_t stream)
{
    ConvArgs p = a;
    p.m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = (p.Cout + HBN - 1) / HBN;
    constexpr size_t stage_b = (size_t)3 * BM * 128 + (size_t)(BM == 256 ? 3 : 2) * H_BSTAGE, ep_b = (size_t)BM * (HBN + 4) * 4;
    constexpr size_t lds = stage_b > ep_b ? stage_b : ep_b;
    static_assert(lds <= 160 * 1024 && (BM == 256 || lds <= 80 * 1024), "one workgroup per CU at 256 pixels, two at 128");
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_conv_igemm_bf16g<BM, PADDED, HAS_X2>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long groups = (p.m_tiles + 7) / 8;
    const long grid = groups * 8 * p.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BM * 2), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// As hvn_launch_conv_bf16 (same operands, same packing, same bits): bm = 256 | 128 pixels x 128 channels per workgroup; no prologue,
// no batched launch, cout >= 128.
int hvn_launch_conv_bf16g(const ConvArgs &a, int bm, hipStream_t stream)
{
    if ((bm != 256 && bm != 128) || a.Cout < 128 || a.pre_s || a.nbatch > 1) return -1;
    if (a.Cin % 32 != 0 || a.Cin <= 0 || a.Cout % 8 != 0) return -1;
    if ((((uintptr_t)a.y) & 15) || ((a.ysn | a.ysy | a.ysx) & 7) || (a.res && ((((uintptr_t)a.res) & 15) || ((a.rsn | a.rsy | a.rsx) & 7)))) return -1;
    if (a.M <= 0 || a.M >= (1L << 31) - 512) return -1;
    const long howo = (long)a.Ho * a.Wo;
    if (howo <= 0) return -1;
    const long ahead = (howo + bm - 2) / howo;       // samples a bm-row tile reaches beyond its first row's (hvn_launch_conv)
    const long span = ahead * a.xsn + (long)(a.H + a.KH) * a.xsy + (long)(a.W + a.KW) * a.xsx;
    if (span < 0 || span * 2 >= (1L << 31)) return -1;
    // the epilogue's 32-bit offsets into y / res, from the sample of the tile's first row
    if ((ahead * a.ysn + (long)(a.Ho + 1) * a.ysy + (long)a.Wo * a.ysx) * 2 >= (1L << 31)) return -1;
    if (a.res && (ahead * a.rsn + (long)(a.Ho + 1) * a.rsy + (long)a.Wo * a.rsx) * 2 >= (1L << 31)) return -1;
    if (a.x2 && (ahead * a.x2sn + (long)a.H * a.x2sy * a.stride2) * 2 >= (1L << 31)) return -1;
    const long kt = (long)a.KH * a.KW * ((a.Cin + HK - 1) / HK) + (a.x2 ? a.Cin2 / HK : 0);
    if ((long)(a.Cout + 128) * kt * HK * 2 >= (1L << 31)) return -1;
    const bool padded = a.pad_t > 0 || a.pad_l > 0 || (a.Ho - 1) * a.stride - a.pad_t + a.KH > a.H ||
                        (a.Wo - 1) * a.stride - a.pad_l + a.KW > a.W;
    if (a.x2) {
        if (padded || a.Cin2 % HK || a.Cin % HK) return -1;
        return bm == 256 ? launch_bf16g<256, false, true>(a, stream) : launch_bf16g<128, false, true>(a, stream);
    }
    if (bm == 256) return padded ? launch_bf16g<256, true, false>(a, stream) : launch_bf16g<256, false, false>(a, stream);
    return padded ? launch_bf16g<128, true, false>(a, stream) : launch_bf16g<128, false, false>(a, stream);
}
