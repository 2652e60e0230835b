// hvn_conv_chain_x3.hip -- hvn_conv_chain.hip's two chained 1x1 convolutions of a pre-activation residual block (a unit's conv3 + residual
// | fused shortcut (+ block-closing BN-ReLU), then the NEXT unit's pre-activation + conv1, in one launch: reference
// /root/reference/models/hovernet/net_utils.py:250-266) with the PRODUCTS of both GEMMs on the gfx950 bf16 matrix pipe from exact three-way
// bf16 splits of the fp32 operands (hvn_conv_x3.hip: x = h + m + l, six or nine exact partial products per product, fp32 accumulation).
//
// Why (round 5): d0's three seams were the largest launches left on the fp32 matrix pipe -- 6.6 ms of a 47 ms step at 84 - 90 TFLOP/s,
// matrix-bound at 60 % of the fp32 peak -- while their compulsory HBM bytes (4.0 / 5.7 / 6.3 GB per seam at batch 32) take ~1.1 ms each.
// Six bf16 MFMAs per product cost 6/16 of the fp32 pipe's time.
//
// Structure = hvn_conv_chain_f32's (same chunk walk, same barriers, register-staged loads under the previous step's MFMAs), with:
//   * weights as the pre-split planes of hvn_conv_x3.hip's packing ([rows][k-step][3][32] bf16), staged into LDS as [plane][row][64 B]
//     with 16-byte piece p of row r at p ^ ((r >> 2) & 3) (conflict-free ds_read_b128 without padding: hvn_conv_x3g.hip);
//   * activations kept fp32 in LDS (GEMM1: the XOR-swizzled k-step tiles; GEMM2: the epilogue tile) and split into their planes at the
//     FRAGMENT READ: a wave owns 32 pixel rows of GEMM1 (and of GEMM2 for cout2 = 64), so nothing is split twice within a chunk.
// Every output element sums the partial products hvn_conv_igemm_x3 would sum, in its order, and the epilogues apply its operations: y and
// t1' are BIT-IDENTICAL to the two bf16x3 CONV launches the op replaces (tests/test_gpu_chain.py).
//
// LDS: GEMM1 staging 2 x (128 x 128 B of A + 12 KB of planes) = 56 KB, aliased by the epilogue / GEMM2-A tile [128][68] floats; the W1'
// chunk's planes behind it: both k-steps of 32 for cout2 = 64 (24 KB), ONE at a time for cout2 = 128 (24 KB; the second waits in registers).
// 80 KB: two workgroups per CU.  Registers: 232 - 244 VGPRs for cout2 = 64; the cout2 = 128 instantiations (one launch per step: d0's last
// seam into d1) sit at the 256 cap with 9 - 27 spilled VGPRs (40 - 112 B of scratch, hipcc -Rpass-analysis=kernel-resource-usage).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CX_BM 128
#define CX_BN 64
#define CX_EP 68                      // epilogue / GEMM2-A tile row pitch (floats)
#define CX_ASTAGE (CX_BM * 128)       // bytes: [128 rows][32 floats]
#define CX_BSTAGE (3 * CX_BN * 64)    // bytes: [3 planes][64 rows][64 B]

static __device__ __forceinline__ f32x4 cx_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
static __device__ __forceinline__ void cx_store(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// hvn_conv_x3.hip:split3 over the 8 k-values a lane feeds one MFMA with
static __device__ __forceinline__ void cx_split(const f32x4 a, const f32x4 b, bf16x8 &h, bf16x8 &m, bf16x8 &l)
{
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hh = (__bf16)x[e];
        const float r = x[e] - (float)hh;
        const __bf16 mm = (__bf16)r;
        h[e] = hh;
        m[e] = mm;
        l[e] = (__bf16)(r - (float)mm);
    }
}

template <int N2, bool HAS_X2, int NTERMS>
__global__ __launch_bounds__(256, 2) void hvn_conv_chain_x3(const ChainArgs p)
{
    constexpr int BM = CX_BM;
    constexpr int PA = BM / 32;                       // A staging passes: 32 rows of 8 float4 per pass
    constexpr int PB = 3;                             // B staging: 64 rows x 12 pieces of 16 B = 768 pieces / 256 threads
    constexpr int PB2 = N2 * 12 / 256;                // W1' chunk, per k-step: 3 | 6
    constexpr bool B2ONE = N2 == 128;                 // W1' chunk: one k-step of 32 in LDS at a time
    constexpr int WAVES_M2 = N2 == 128 ? 2 : 4, WAVES_N2 = 4 / WAVES_M2;
    constexpr int WM2 = BM / WAVES_M2, WN2 = N2 / WAVES_N2;
    constexpr int TM2 = WM2 / 32, TN2 = WN2 / 32;     // 1 x 2 | 2 x 2
    constexpr int B2PLANE = N2 * 64;                  // bytes of one plane of a W1' k-step
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NTERMS == 9 || NTERMS == 6, "nine exact partial products, or the six that carry > 2^-24 of the product");
    extern __shared__ __attribute__((aligned(16))) unsigned char cs[];
    unsigned char *As = cs;                                   // [2][128][128 B]
    unsigned char *Bs = cs + 2 * CX_ASTAGE;                   // [2][3][64][64 B]
    float *ep = (float *)cs;                                  // [128][68], aliases As / Bs
    unsigned char *B2s = cs + 2 * (CX_ASTAGE + CX_BSTAGE);    // [2 | 1][3][N2][64 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned M = (unsigned)p.M;
    const unsigned m0 = blockIdx.x * (unsigned)BM;
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const unsigned n_blk = m0 / HoWo;

    // ---- A staging coordinates (one 16-byte piece of a 32-float k-slab row per thread and pass) -------------------------
    const int srow = tid >> 3;
    const int scol = (tid & 7) * 4;
    const int lcol = ((tid & 7) ^ ((srow >> 1) & 7)) * 4;
    unsigned a_voff[PA], a2_voff[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const unsigned m = m0 + srow + 32 * j;
        const bool ok = m < M;
        const unsigned mm = ok ? m : m0;
        const unsigned n = mm / HoWo;
        const unsigned rem = mm - n * HoWo;
        const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
        a_voff[j] = ok ? (unsigned)(((long)(n - n_blk) * p.xsn + (long)oy * p.xsy + (long)ox * p.xsx + scol) * 4) : OOB;
        a2_voff[j] = (ok && HAS_X2) ? (unsigned)(((long)(n - n_blk) * p.x2sn + (long)(oy * p.stride2) * p.x2sy + (long)(ox * p.stride2) * p.x2sx + scol) * 4) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (long)n_blk * p.xsn), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x2 =
        __builtin_amdgcn_make_buffer_rsrc((void *)(HAS_X2 ? p.x2 + (long)n_blk * p.x2sn : p.x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void *)p.w2, 0, 0x7fffffff, 0x00020000);
    const int KT1 = p.K1 / 32;
    const int KT = KT1 + (HAS_X2 ? p.K1b / 32 : 0);
    const int NC = p.C / CX_BN;
    const int KT2 = p.C / 32;                                 // k-steps of the second GEMM over all chunks (row length of W1')
    // ---- weight staging: 16-byte piece c = tid + 256 j of the [rows][12] pieces of one k-step; LDS slot = [plane][row][piece ^ ((row >> 2) & 3)]
    unsigned w_voff[PB], w_lds[PB], w2_voff[PB2], w2_lds[PB2];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int c = tid + 256 * j;
        const int row = c / 12, pc = c - row * 12;
        w_voff[j] = (unsigned)(((long)row * KT * 192) + pc * 16);
        w_lds[j] = (unsigned)((pc >> 2) * (CX_BN * 64) + row * 64 + (((pc & 3) ^ ((row >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int j = 0; j < PB2; ++j) {
        const int c = tid + 256 * j;
        const int row = c / 12, pc = c - row * 12;
        w2_voff[j] = (unsigned)(((long)row * KT2 * 192) + pc * 16);
        w2_lds[j] = (unsigned)((pc >> 2) * B2PLANE + row * 64 + (((pc & 3) ^ ((row >> 2) & 3)) << 4));
    }

    // ---- epilogue coordinates: thread = (16-byte column piece, rows erow0 + 16 it) -------------------------------------
    const int ecol = (tid & 15) * 4;
    const int erow0 = tid >> 4;
    constexpr int NIT = BM / 16;
    unsigned y_voff[NIT];      // the residual view has the output's strides (validated by the launcher): same offsets, other base
    const bool has_res = p.res != nullptr;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const unsigned m = m0 + erow0 + 16 * it;
        const bool ok = m < M;
        const unsigned mm = ok ? m : m0;
        const unsigned n = mm / HoWo;
        const unsigned rem = mm - n * HoWo;
        const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
        y_voff[it] = ok ? (unsigned)(((long)(n - n_blk) * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + ecol) * 4) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void *)(p.y + (long)n_blk * p.ysn), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r =
        __builtin_amdgcn_make_buffer_rsrc((void *)(has_res ? p.res + (long)n_blk * p.rsn : p.x), 0, 0x7fffffff, 0x00020000);
    const bool has_post = p.post_s != nullptr, has_pre = p.pre_s != nullptr;
    const float post_lo = has_post ? 0.f : -__builtin_inff();
    const float pre_lo = has_pre ? 0.f : -__builtin_inff();

    struct Stage {
        f32x4 ra[PA];
        u32x4 rb[PB];
    };
    Stage st;
    auto load1 = [&](int c, int kt) {   // raw loads of GEMM1 k-step kt of chunk c (nothing waits here)
        const bool second = HAS_X2 && kt >= KT1;                  // uniform
        const int a_soff = (second ? kt - KT1 : kt) * 128;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            if constexpr (HAS_X2)
                st.ra[j] = cx_load(second ? rsrc_x2 : rsrc_x, second ? a2_voff[j] : a_voff[j], a_soff);
            else
                st.ra[j] = cx_load(rsrc_x, a_voff[j], a_soff);
        }
        const int w_soff = (c * CX_BN * KT + kt) * 192;
#pragma unroll
        for (int j = 0; j < PB; ++j) st.rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w1, w_voff[j], w_soff, 0);
    };
    auto store1 = [&](int buf) {
        float *a = (float *)(As + buf * CX_ASTAGE);
        unsigned char *b = Bs + buf * CX_BSTAGE;
#pragma unroll
        for (int j = 0; j < PA; ++j) *(f32x4 *)(a + (srow + 32 * j) * 32 + lcol) = st.ra[j];
#pragma unroll
        for (int j = 0; j < PB; ++j) *(u32x4 *)(b + w_lds[j]) = st.rb[j];
    };

    f32x16 acc1[2];
    f32x16 acc2[TM2][TN2];
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    const int akey = (l31 >> 1) & 7, bkey = (l31 >> 2) & 3;
    const int wm2 = wave / WAVES_N2, wn2 = wave % WAVES_N2;

    // the 4 | 6 | 9 ... partial products of one 16-deep slice for one accumulator, smallest first (hvn_conv_igemm_x3's order)
    auto mac = [&](f32x16 &acc, const bf16x8 (&fa)[3], const bf16x8 (&fb)[3], int s_only) {
#pragma unroll
        for (int pa = 2; pa >= 0; --pa) {
            const int pb = s_only - pa;
            if (pb < 0 || pb > 2) continue;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa], fb[pb], acc, 0, 0, 0);
        }
    };
    auto mma1 = [&](int buf) {          // wave = 32 pixels x the chunk's 64 channels
        const unsigned char *a = As + buf * CX_ASTAGE + (wave * 32 + l31) * 128;
        const unsigned char *b = Bs + buf * CX_BSTAGE + l31 * 64;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 v0 = *(const f32x4 *)(a + (((4 * q + 2 * lh) ^ akey) << 4));
            const f32x4 v1 = *(const f32x4 *)(a + (((4 * q + 2 * lh + 1) ^ akey) << 4));
            bf16x8 fa[3], fb[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fb[j][pl] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(b + pl * (CX_BN * 64) + j * 32 * 64 + (((2 * q + lh) ^ bkey) << 4)));
            cx_split(v0, v1, fa[0], fa[1], fa[2]);
#pragma unroll
            for (int s = 4; s >= 0; --s) {
                if (NTERMS == 6 && s > 2) continue;
                // hvn_conv_igemm_x3 walks (s, pa) outside and the accumulators inside: per accumulator the order is (s, pa) -- kept
#pragma unroll
                for (int j = 0; j < 2; ++j) mac(acc1[j], fa, fb[j], s);
            }
        }
    };
    auto mma2 = [&](int ks) {           // A = the activated chunk in the epilogue tile (fp32), B = W1' chunk planes: k-step ks (32 channels) of two
        const float *a = ep + (wm2 * WM2 + l31) * CX_EP + ks * 32 + 8 * lh;
        const unsigned char *b = B2s + (B2ONE ? 0 : ks) * (3 * B2PLANE) + (wn2 * WN2 + l31) * 64;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 fa[TM2][3], fb[TN2][3];
#pragma unroll
            for (int i = 0; i < TM2; ++i) {
                const f32x4 v0 = *(const f32x4 *)(a + i * 32 * CX_EP + q * 16);
                const f32x4 v1 = *(const f32x4 *)(a + i * 32 * CX_EP + q * 16 + 4);
                cx_split(v0, v1, fa[i][0], fa[i][1], fa[i][2]);
            }
#pragma unroll
            for (int j = 0; j < TN2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fb[j][pl] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(b + pl * B2PLANE + j * 32 * 64 + (((2 * q + lh) ^ bkey) << 4)));
#pragma unroll
            for (int s = 4; s >= 0; --s) {
                if (NTERMS == 6 && s > 2) continue;
#pragma unroll
                for (int i = 0; i < TM2; ++i)
#pragma unroll
                    for (int j = 0; j < TN2; ++j) mac(acc2[i][j], fa[i], fb[j], s);
            }
        }
    };

    load1(0, 0);
    for (int c = 0; c < NC; ++c) {
        // stage of this chunk's first k-step (loaded during the previous chunk's last one).  FIRST thing of the chunk: the wait
        // for it also drains the previous chunk's y stores (one vmcnt for loads and stores), which have had all of GEMM2 to
        // retire -- nothing else may be in flight yet or the wait would include it.
        store1(0);
        // W1' chunk (both k-steps of 32): loaded under the first GEMM1 step, parked in its own LDS region (free since the barrier
        // behind the previous GEMM2) right after it; with one k-step of LDS (cout2 = 128) the second stays in registers until GEMM2
        u32x4 rb2[2][PB2];
#pragma unroll
        for (int ks = 0; ks < (B2ONE ? 1 : 2); ++ks)      // (cout2 = 128: the second k-step is loaded in epilogue 1 -- 24 registers less across GEMM1)
#pragma unroll
            for (int j = 0; j < PB2; ++j) rb2[ks][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w2, w2_voff[j], (2 * c + ks) * 192, 0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
        // first k-step (KT >= 2: validated by the launcher)
        load1(c, 1);
        mma1(0);
#pragma unroll
        for (int ks = 0; ks < (B2ONE ? 1 : 2); ++ks)
#pragma unroll
            for (int j = 0; j < PB2; ++j) *(u32x4 *)(B2s + ks * (3 * B2PLANE) + w2_lds[j]) = rb2[ks][j];
        store1(1);
        __syncthreads();
        for (int kt = 1; kt + 1 < KT; ++kt) {
            load1(c, kt + 1);
            mma1(kt & 1);
            store1((kt + 1) & 1);
            __syncthreads();
        }
        // last k-step: the residual tile, then the first stage of the next chunk, fly under it
        f32x4 rres[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            rres[it] = has_res ? cx_load(rsrc_r, y_voff[it], c * (CX_BN * 4)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c + 1 < NC) load1(c + 1, 0);
        mma1((KT - 1) & 1);
        __syncthreads();               // every wave is done reading the staging buffers: the tile may overwrite them
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                ep[row * CX_EP + j * 32 + l31] = acc1[j][r];
            }
        __syncthreads();
        // ---- epilogue 1: + residual, block-closing BN-ReLU, store y, next unit's pre-activation back into the tile -----
        {
            const int co = c * CX_BN + ecol;
            f32x4 qs = {1.f, 1.f, 1.f, 1.f}, qb = {0.f, 0.f, 0.f, 0.f}, ps = qs, pb = qb;
            if (has_post) {
                qs = *(const f32x4 *)(p.post_s + co);
                qb = *(const f32x4 *)(p.post_b + co);
            }
            if (has_pre) {
                ps = *(const f32x4 *)(p.pre_s + co);
                pb = *(const f32x4 *)(p.pre_b + co);
            }
            f32x4 vout[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float *e = ep + (erow0 + 16 * it) * CX_EP + ecol;
                f32x4 v = *(const f32x4 *)e;
                v += rres[it];
                v.x = fmaxf(fmaf(v.x, qs.x, qb.x), post_lo);
                v.y = fmaxf(fmaf(v.y, qs.y, qb.y), post_lo);
                v.z = fmaxf(fmaf(v.z, qs.z, qb.z), post_lo);
                v.w = fmaxf(fmaf(v.w, qs.w, qb.w), post_lo);
                vout[it] = v;
                f32x4 a;
                a.x = fmaxf(fmaf(v.x, ps.x, pb.x), pre_lo);
                a.y = fmaxf(fmaf(v.y, ps.y, pb.y), pre_lo);
                a.z = fmaxf(fmaf(v.z, ps.z, pb.z), pre_lo);
                a.w = fmaxf(fmaf(v.w, ps.w, pb.w), pre_lo);
                *(f32x4 *)e = a;
            }
            // the stores leave back to back, after every value is final (see hvn_conv.hip: one vmcnt for loads and stores)
#pragma unroll
            for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(vout[it]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (B2ONE) {     // W1' k-step 1: issued AHEAD of the y stores (one in-order vmcnt: waiting for it later leaves the stores in flight)
#pragma unroll
                for (int j = 0; j < PB2; ++j) rb2[1][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w2, w2_voff[j], (2 * c + 1) * 192, 0);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) cx_store(vout[it], rsrc_y, y_voff[it], c * (CX_BN * 4));
        }
        __syncthreads();
        mma2(0);
        if constexpr (B2ONE) {
            __syncthreads();           // every wave is done with k-step 0 of W1': its LDS region takes k-step 1
#pragma unroll
            for (int j = 0; j < PB2; ++j) *(u32x4 *)(B2s + w2_lds[j]) = rb2[1][j];
            __syncthreads();
        }
        mma2(1);
        __syncthreads();               // tile and W1' chunk are free again
    }

    // ---- epilogue 2: t1' = relu(acc2 + b2), 64 output channels at a time through the tile ----------------------------
    unsigned y2_voff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const unsigned m = m0 + erow0 + 16 * it;
        const bool ok = m < M;
        const unsigned mm = ok ? m : m0;
        const unsigned n = mm / HoWo;
        const unsigned rem = mm - n * HoWo;
        const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
        y2_voff[it] = ok ? (unsigned)(((long)(n - n_blk) * p.y2sn + (long)oy * p.y2sy + (long)ox * p.y2sx + ecol) * 4) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsrc_y2 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.y2 + (long)n_blk * p.y2sn), 0, 0x7fffffff, 0x00020000);
    const float relu_lo = p.relu2 ? 0.f : -__builtin_inff();
#pragma unroll
    for (int h = 0; h < N2 / 64; ++h) {
#pragma unroll
        for (int j = 0; j < TN2; ++j) {
            const int col0 = wn2 * WN2 + j * 32;        // this wave's 32-column tile j: does it belong to the 64-channel half h?
            if (col0 / 64 == h) {
#pragma unroll
                for (int i = 0; i < TM2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm2 * WM2 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        ep[row * CX_EP + (col0 & 63) + l31] = acc2[i][j][r];
                    }
            }
        }
        __syncthreads();
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (p.bias2) bias = *(const f32x4 *)(p.bias2 + h * 64 + ecol);
        f32x4 vout[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f32x4 v = *(const f32x4 *)(ep + (erow0 + 16 * it) * CX_EP + ecol);
            v.x = fmaxf(v.x + bias.x, relu_lo);
            v.y = fmaxf(v.y + bias.y, relu_lo);
            v.z = fmaxf(v.z + bias.z, relu_lo);
            v.w = fmaxf(v.w + bias.w, relu_lo);
            vout[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) cx_store(vout[it], rsrc_y2, y2_voff[it], h * 256);
        if (h + 1 < N2 / 64) __syncthreads();
    }
}

template <int N2, bool HAS_X2, int NTERMS>
static int launch_chain_x3(const ChainArgs &a, hipStream_t stream)
{
    constexpr size_t lds = (size_t)2 * (CX_ASTAGE + CX_BSTAGE) + (size_t)(N2 == 128 ? 1 : 2) * 3 * N2 * 64;
    static_assert((size_t)CX_BM * CX_EP * 4 <= (size_t)2 * (CX_ASTAGE + CX_BSTAGE), "the epilogue tile must fit the staging buffers it aliases");
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_conv_chain_x3<N2, HAS_X2, NTERMS>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long grid = (a.M + CX_BM - 1) / CX_BM;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ChainArgs as hvn_launch_conv_chain, except: w1 / w2 = the bf16 planes of the fp32 packings ([rows][k-step][3][32] bf16: hvn_conv_x3.hip),
// 128 pixels per workgroup, terms = 9 | 6 partial products per product.
int hvn_launch_conv_chain_x3(const ChainArgs &a, int terms, hipStream_t stream)
{
    if (!hvn_chain_supported(a.C, a.N2) || a.K1 <= 0 || a.K1 % 32 || a.K1 + a.K1b < 64 || (a.x2 && (a.K1b <= 0 || a.K1b % 32))) return -1;
    if (a.M <= 0 || a.M >= (1L << 31) - 256) return -1;
    const long px = (long)a.Ho * a.Wo;
    if (px <= 0) return -1;
    const long ns = 128 / px + 2;
    const long spans[5] = {ns * a.xsn, a.x2 ? ns * a.x2sn : 0, ns * a.ysn, a.res ? ns * a.rsn : 0, ns * a.y2sn};
    for (long s : spans)
        if (s < 0 || s * 4 >= (1L << 31)) return -1;
    if ((long)(a.C + 64) * ((a.K1 + a.K1b) / 32) * 192 >= (1L << 31) || (long)(a.N2 + 64) * (a.C / 32) * 192 >= (1L << 31)) return -1;
    const bool nine = terms != 6;
    if (a.N2 == 64) {
        if (a.x2) return nine ? launch_chain_x3<64, true, 9>(a, stream) : launch_chain_x3<64, true, 6>(a, stream);
        return nine ? launch_chain_x3<64, false, 9>(a, stream) : launch_chain_x3<64, false, 6>(a, stream);
    }
    if (a.x2) return nine ? launch_chain_x3<128, true, 9>(a, stream) : launch_chain_x3<128, true, 6>(a, stream);
    return nine ? launch_chain_x3<128, false, 9>(a, stream) : launch_chain_x3<128, false, 6>(a, stream);
}
