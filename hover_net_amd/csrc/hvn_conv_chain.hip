// hvn_conv_chain.hip -- two chained 1x1 convolutions of a pre-activation residual block in ONE launch (gfx950, fp32 MFMA).
//
// Reference: /root/reference/models/hovernet/net_utils.py:250-266 (ResidualBlock.forward).  Unit i ends with
//     y = conv3(t2) + shortcut                      (1x1, C/4 -> C; unit 0: shortcut = 1x1 conv of the block input)
// and unit i+1 (or unit 0 of the next block, after blk_bna) begins with
//     t1' = relu(bn(conv1(relu(bn_pre(y)))))        (1x1, C -> C/4)
// Both are per-pixel: the second needs no neighbour of the first, so a workgroup that owns ALL C channels of its 128 pixels can
// run conv1 straight on the y values it has just produced.  What this removes per pixel and seam: one read of y (4C bytes; the
// dominant term of the HBM-bound d0 / d1 layers: 4 KB -> 3 KB per pixel and unit in d0), one launch, and one prologue-k-loop-
// epilogue round trip of short-K workgroups (profiles/r02_experiments.md section 7).  No halo, so it composes with the Winograd
// form of conv2 in d1..d3.
//
// Work of one 256-thread workgroup (128 pixels), for each 64-channel chunk c of C:
//   GEMM1   acc1[128 px][64]  = sum_k x[px][k] W3[64c + j][k]  (+ the shortcut's channels appended to k)      K1 = C/4 (+ Cin2)
//   epi 1   v = acc1 + res;  v = relu(v*qs + qb) if the block closes here;  y[px][64c..] = v   (16 B per lane, 256 B per row)
//           a = relu(v*ps + pb) (next unit's pre-activation; identity after a block-closing BN-ReLU)  -> stays in LDS
//   GEMM2   acc2[128 px][N2] += sum_{j<64} a[px][j] W1'[n][64c + j]                                     N2 = C'/4 in {64, 128}
// and after the last chunk   t1'[px][n] = relu(acc2 + b2).
// Every output element sums its products in exactly the order of hvn_conv_igemm_f32 (32-channel slabs in order, the same k
// permutation inside a slab, one accumulator), so y and t1' are BIT-IDENTICAL to the two separate launches (tests/test_gpu_chain.py).
//
// LDS (floats): GEMM1 staging [2][128][32] + [2][64][32] (48 KB, XOR-swizzled rows as in hvn_conv.hip); the epilogue tile
// [128][68] aliases it and doubles as GEMM2's A operand (row pitch 68: conflict-free ds_read_b128 for 16 consecutive rows);
// W1' chunk [2][N2][32] behind it.  64 KB (N2 = 64) / 80 KB (N2 = 128): two workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CH_BM 128
#define CH_BN 64
#define CH_EP 68   // epilogue / GEMM2-A tile row pitch (floats)

static __device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
static __device__ __forceinline__ void buf_store(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// BM = pixels per workgroup: 128 (two workgroups per CU) or 64 (W1' staged one k-step at a time: 40 / 48 KB of LDS, <= 168 VGPRs, three
// workgroups per CU -- for the seams that are bound by exposed load latency rather than by bytes; chosen per launch by the engine's
// timing pass, `tile_n` of the op).  Every output element is summed in the same order for both.
template <int BM, int N2, bool HAS_X2>
__global__ __launch_bounds__(256, BM == 64 ? 3 : 2) void hvn_conv_chain_f32(const ChainArgs p)
{
    constexpr int PA = BM / 32, PB = CH_BN / 32, PB2 = N2 / 32;
    constexpr int W1M = BM / 32 >= 4 ? 4 : BM / 32, W1N = 4 / W1M;          // GEMM1: waves over pixels x the chunk's 64 channels
    constexpr int TN1 = CH_BN / W1N / 32;
    constexpr int WM2_MAX = N2 >= 128 ? 2 : 4;
    constexpr int WAVES_M2 = BM / 32 < WM2_MAX ? BM / 32 : WM2_MAX, WAVES_N2 = 4 / WAVES_M2;
    constexpr int WM2 = BM / WAVES_M2, WN2 = N2 / WAVES_N2;
    constexpr int TM2 = WM2 / 32, TN2 = WN2 / 32;
    constexpr bool B2ONE = BM == 64;                                       // W1' chunk: one k-step of 32 in LDS at a time
    constexpr unsigned OOB = 0x80000000u;
    static_assert(W1M * W1N == 4 && TN1 >= 1 && TM2 >= 1 && TN2 >= 1, "256 threads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                              // [2][BM][32]
    float *Bs = smem + 2 * BM * 32;                // [2][64][32]
    float *ep = smem;                              // [BM][68], aliases As / Bs
    float *B2s = smem + 2 * (BM + CH_BN) * 32;     // [2 | 1][N2][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned M = (unsigned)p.M;
    const unsigned m0 = blockIdx.x * (unsigned)BM;
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const unsigned n_blk = m0 / HoWo;

    // ---- staging coordinates (one 16-byte piece of a 32-float k-slab row per thread and pass) -------------------------
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
    const int Ktot = KT * 32;
    const int NC = p.C / CH_BN;
    unsigned w_voff[PB], w2_voff[PB2];
#pragma unroll
    for (int j = 0; j < PB; ++j) w_voff[j] = (unsigned)(((srow + 32 * j) * Ktot + scol) * 4);
#pragma unroll
    for (int j = 0; j < PB2; ++j) w2_voff[j] = (unsigned)(((srow + 32 * j) * p.C + scol) * 4);

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
        f32x4 ra[PA], rb[PB];
    };
    Stage st;
    auto load1 = [&](int c, int kt) {   // raw loads of GEMM1 k-step kt of chunk c (nothing waits here)
        const bool second = HAS_X2 && kt >= KT1;                  // uniform
        const int a_soff = (second ? kt - KT1 : kt) * 128;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            if constexpr (HAS_X2)
                st.ra[j] = buf_load(second ? rsrc_x2 : rsrc_x, second ? a2_voff[j] : a_voff[j], a_soff);
            else
                st.ra[j] = buf_load(rsrc_x, a_voff[j], a_soff);
        }
        const int w_soff = (c * CH_BN * Ktot + kt * 32) * 4;
#pragma unroll
        for (int j = 0; j < PB; ++j) st.rb[j] = buf_load(rsrc_w1, w_voff[j], w_soff);
    };
    auto store1 = [&](int buf) {
        float *a = As + buf * BM * 32;
        float *b = Bs + buf * CH_BN * 32;
#pragma unroll
        for (int j = 0; j < PA; ++j) *(f32x4 *)(a + (srow + 32 * j) * 32 + lcol) = st.ra[j];
#pragma unroll
        for (int j = 0; j < PB; ++j) *(f32x4 *)(b + (srow + 32 * j) * 32 + lcol) = st.rb[j];
    };

    f32x16 acc1[TN1];
    f32x16 acc2[TM2][TN2];
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    const int key = (l31 >> 1) & 7;
    const int wm2 = wave / WAVES_N2, wn2 = wave % WAVES_N2;
    const int w1m = wave / W1N, w1n = wave % W1N;

    auto mma1 = [&](int buf) {          // wave = 32 pixels x (64 / W1N) channels of the chunk
        const float *a = As + buf * BM * 32 + (w1m * 32 + l31) * 32;
        const float *b = Bs + buf * CH_BN * 32 + (w1n * TN1 * 32 + l31) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = ((2 * q + lh) ^ key) * 4;
            const f32x4 fa = *(const f32x4 *)(a + off);
            f32x4 fb[TN1];
#pragma unroll
            for (int j = 0; j < TN1; ++j) fb[j] = *(const f32x4 *)(b + j * 32 * 32 + off);
#pragma unroll
            for (int j = 0; j < TN1; ++j) {
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[j].x, acc1[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[j].y, acc1[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[j].z, acc1[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[j].w, acc1[j], 0, 0, 0);
            }
        }
    };
    auto mma2 = [&](int ks) {           // A = the activated chunk in the epilogue tile, B = W1' chunk: k-step ks (32 channels) of two
        const float *a = ep + (wm2 * WM2 + l31) * CH_EP;
        const float *b = B2s + (B2ONE ? 0 : ks) * N2 * 32 + (wn2 * WN2 + l31) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 fa[TM2], fb[TN2];
            const int off = ((2 * q + lh) ^ key) * 4;
#pragma unroll
            for (int i = 0; i < TM2; ++i) fa[i] = *(const f32x4 *)(a + i * 32 * CH_EP + ks * 32 + q * 8 + 4 * lh);
#pragma unroll
            for (int j = 0; j < TN2; ++j) fb[j] = *(const f32x4 *)(b + j * 32 * 32 + off);
#pragma unroll
            for (int i = 0; i < TM2; ++i)
#pragma unroll
                for (int j = 0; j < TN2; ++j) {
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc2[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc2[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc2[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc2[i][j], 0, 0, 0);
                }
        }
    };

    // diagnosis (HVN_CHAIN_TRACE): cycle stamps of one steady-state chunk (the second, or the only one) of every workgroup
    unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int c_dbg = p.dbg ? (NC > 1 ? 1 : 0) : -1;
#define CH_STAMP(i) if (c == c_dbg) ts[i] = __builtin_readcyclecounter()
    if (p.dbg) ts[0] = __builtin_readcyclecounter();
    load1(0, 0);
    for (int c = 0; c < NC; ++c) {
        CH_STAMP(1);
        // stage of this chunk's first k-step (loaded during the previous chunk's last one).  FIRST thing of the chunk: the wait
        // for it also drains the previous chunk's y stores (one vmcnt for loads and stores), which have had all of GEMM2 to
        // retire -- nothing else may be in flight yet or the wait would include it.
        store1(0);
        // W1' chunk (both k-steps of 32): loaded under the first GEMM1 step, parked in its own LDS region (free since the barrier
        // behind the previous GEMM2) right after it; with one k-step of LDS (BM = 64) the second stays in registers until GEMM2
        f32x4 rb2[2][PB2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < PB2; ++j) rb2[ks][j] = buf_load(rsrc_w2, w2_voff[j], ((2 * c + ks) * 32) * 4);
        __syncthreads();
        CH_STAMP(2);
#pragma unroll
        for (int j = 0; j < TN1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
        // first k-step (KT >= 2: validated by the launcher)
        load1(c, 1);
        mma1(0);
#pragma unroll
        for (int ks = 0; ks < (B2ONE ? 1 : 2); ++ks)
#pragma unroll
            for (int j = 0; j < PB2; ++j) *(f32x4 *)(B2s + ks * N2 * 32 + (srow + 32 * j) * 32 + lcol) = rb2[ks][j];
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
            rres[it] = has_res ? buf_load(rsrc_r, y_voff[it], c * (CH_BN * 4)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c + 1 < NC) load1(c + 1, 0);
        mma1((KT - 1) & 1);
        CH_STAMP(3);
        __syncthreads();               // every wave is done reading the staging buffers: the tile may overwrite them
#pragma unroll
        for (int j = 0; j < TN1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = w1m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                ep[row * CH_EP + (w1n * TN1 + j) * 32 + l31] = acc1[j][r];
            }
        __syncthreads();
        CH_STAMP(4);
        // ---- epilogue 1: + residual, block-closing BN-ReLU, store y, next unit's pre-activation back into the tile -----
        {
            const int co = c * CH_BN + ecol;
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
                float *e = ep + (erow0 + 16 * it) * CH_EP + ecol;
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
#pragma unroll
            for (int it = 0; it < NIT; ++it) buf_store(vout[it], rsrc_y, y_voff[it], c * (CH_BN * 4));
        }
        CH_STAMP(5);
        __syncthreads();
        CH_STAMP(6);
        mma2(0);
        if constexpr (B2ONE) {
            __syncthreads();           // every wave is done with k-step 0 of W1': its LDS tile takes k-step 1
#pragma unroll
            for (int j = 0; j < PB2; ++j) *(f32x4 *)(B2s + (srow + 32 * j) * 32 + lcol) = rb2[1][j];
            __syncthreads();
        }
        mma2(1);
        CH_STAMP(7);
        __syncthreads();               // tile and W1' chunk are free again
    }
    if (p.dbg) ts[8] = __builtin_readcyclecounter();

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
                        ep[row * CH_EP + (col0 & 63) + l31] = acc2[i][j][r];
                    }
            }
        }
        __syncthreads();
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (p.bias2) bias = *(const f32x4 *)(p.bias2 + h * 64 + ecol);
        f32x4 vout[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f32x4 v = *(const f32x4 *)(ep + (erow0 + 16 * it) * CH_EP + ecol);
            v.x = fmaxf(v.x + bias.x, relu_lo);
            v.y = fmaxf(v.y + bias.y, relu_lo);
            v.z = fmaxf(v.z + bias.z, relu_lo);
            v.w = fmaxf(v.w + bias.w, relu_lo);
            vout[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) buf_store(vout[it], rsrc_y2, y2_voff[it], h * 256);
        if (h + 1 < N2 / 64) __syncthreads();
    }
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[9] = __builtin_readcyclecounter();
        unsigned long long *d = p.dbg + 10ull * blockIdx.x;
#pragma unroll
        for (int i = 0; i < 10; ++i) d[i] = ts[i];
    }
#undef CH_STAMP
}

template <int BM, int N2, bool HAS_X2>
static int launch_chain(const ChainArgs &a, hipStream_t stream)
{
    constexpr size_t lds = (size_t)(2 * (BM + CH_BN) * 32 + (BM == 64 ? 1 : 2) * N2 * 32) * sizeof(float);
    static_assert((size_t)BM * CH_EP <= (size_t)2 * (BM + CH_BN) * 32, "the epilogue tile must fit the staging buffers it aliases");
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_conv_chain_f32<BM, N2, HAS_X2>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long grid = (a.M + BM - 1) / BM;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    static unsigned long long *dbg_buf = nullptr;
    static int dbg_on = -1;
    if (dbg_on < 0) {
        const char *e = getenv("HVN_CHAIN_TRACE");   // path of a file to dump the per-workgroup stamps of the LAST launch into
        dbg_on = e ? 1 : 0;
        if (dbg_on) hipMalloc(&dbg_buf, 10 * 8 * (size_t)(1 << 20));
    }
    ChainArgs p = a;
    p.dbg = (dbg_on && grid <= (1 << 20)) ? dbg_buf : nullptr;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
    if (p.dbg) {   // experiment mode only: synchronous dump
        hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(10 * (size_t)grid);
        hipMemcpy(h.data(), dbg_buf, 8 * h.size(), hipMemcpyDeviceToHost);
        if (FILE *f = fopen(getenv("HVN_CHAIN_TRACE"), "wb")) {
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hvn_chain_supported(int c, int n2) { return c > 0 && c % CH_BN == 0 && (n2 == 64 || n2 == 128); }

int hvn_launch_conv_chain(const ChainArgs &a, hipStream_t stream)
{
    if (!hvn_chain_supported(a.C, a.N2) || a.K1 <= 0 || a.K1 % 32 || a.K1 + a.K1b < 64 || (a.x2 && (a.K1b <= 0 || a.K1b % 32))) return -1;
    if (a.M <= 0 || a.M >= (1L << 31) - 256) return -1;
    // 32-bit per-thread byte offsets below 2^31 (the top bit marks rows past the end): a tile of `bm` pixels spans at most
    // bm / (Ho * Wo) + 2 samples (two when a sample holds at least a tile's worth of pixels, as in every HoVer-Net plan)
    const long px = (long)a.Ho * a.Wo;
    if (px <= 0) return -1;
    const long ns = (a.bm == 64 ? 64 : 128) / px + 2;
    const long spans[5] = {ns * a.xsn, a.x2 ? ns * a.x2sn : 0, ns * a.ysn, a.res ? ns * a.rsn : 0, ns * a.y2sn};
    for (long s : spans)
        if (s < 0 || s * 4 >= (1L << 31)) return -1;
    if ((long)(a.C + 64) * (a.K1 + a.K1b) * 4 >= (1L << 31) || (long)(a.N2 + 64) * a.C * 4 >= (1L << 31)) return -1;
    const bool small = a.bm == 64;          // pixels per workgroup: 128 (default) or 64
    if (a.N2 == 64) {
        if (small) return a.x2 ? launch_chain<64, 64, true>(a, stream) : launch_chain<64, 64, false>(a, stream);
        return a.x2 ? launch_chain<128, 64, true>(a, stream) : launch_chain<128, 64, false>(a, stream);
    }
    if (small) return a.x2 ? launch_chain<64, 128, true>(a, stream) : launch_chain<64, 128, false>(a, stream);
    return a.x2 ? launch_chain<128, 128, true>(a, stream) : launch_chain<128, 128, false>(a, stream);
}
