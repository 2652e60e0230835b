// hvn_conv_chain_x3r.hip -- hvn_conv_chain_x3.hip's chained pair of 1x1 convolutions (a unit's conv3 + residual | fused shortcut
// (+ block-closing BN-ReLU), then the NEXT unit's pre-activation + conv1, both GEMMs' products on the bf16 matrix pipe from exact
// three-way bf16 splits; reference /root/reference/models/hovernet/net_utils.py:250-266) rebuilt around what bounds it: LATENCY.
//
// Why (round 5): d0's seams move 4 - 6 GB per launch (compulsory: ~1.1 ms at 5 TB/s) and took 1.5 - 2.0 ms with the matrix pipe 23 % busy.
// hvn_conv_chain_x3 stages every operand through registers and separates its phases with __syncthreads(), which drains the vector-memory
// counter: the residual tile, issued one short GEMM step before the barrier that precedes its use, is waited for in full at that
// barrier, as are the weight chunks -- ~3 exposed round trips per 64-channel chunk with two workgroups per CU to cover them.  Here:
//   * conv3's INPUT tile (128 pixels x K1 = 64 (+ 64 of the shortcut's input) channels) is loaded ONCE per workgroup, straight from
//     global memory into the MFMA fragment layout, split into its bf16 planes once and kept in registers for all chunks (48 | 96 VGPRs;
//     hvn_conv_chain_x3 re-staged and re-split it per chunk);
//   * the weight chunks (W1: 32 output channels x all k-steps; W1': the matching k-step of the second GEMM, double-buffered) arrive by
//     LDS-DMA (`buffer_load_dwordx4 ... lds`, source-side XOR swizzle: hvn_conv_x3g.hip), issued a whole chunk ahead;
//   * the residual chunk is loaded a whole chunk ahead into registers (16 VGPRs at 32 channels per chunk);
//   * barriers are raw s_barrier with COUNTED s_waitcnt: nothing younger than what a phase needs is waited for, so the residual stream
//     and the y stores stay in flight across all of them.  Three barriers per chunk.
// Chunks are 32 channels of conv3's output (hvn_conv_chain_x3: 64): the epilogue / GEMM2-A tile is [128][32] floats (16 KB, 16-byte
// pieces XOR-swizzled like hvn_conv_x3g.hip's activation rows), LDS 52 - 76 KB, two workgroups per CU.
// Every output element sums the partial products hvn_conv_igemm_x3 would sum, in its order (k-steps ascending, then the partial
// products smallest first), and the epilogues apply its operations: y and t1' are BIT-IDENTICAL to hvn_conv_chain_x3's and to the two
// bf16x3 CONV launches the op replaces (tests/test_gpu_chain.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CR_BM 128
#define CR_CB 32                      // conv3 output channels per chunk

typedef __attribute__((address_space(3))) void *cr_lds_ptr_t;

// one LDS-DMA instruction (hvn_conv_x3g.hip:dma16; the builtin exists in the device pass only)
static __device__ __forceinline__ void cr_dma16(__amdgpu_buffer_rsrc_t rsrc, cr_lds_ptr_t dst, unsigned voff, int soff)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
#endif
}
static __device__ __forceinline__ f32x4 cr_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
static __device__ __forceinline__ void cr_store(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
// hvn_conv_x3.hip:split3 over the 8 k-values a lane feeds one MFMA with
static __device__ __forceinline__ void cr_split(const f32x4 a, const f32x4 b, bf16x8 &h, bf16x8 &m, bf16x8 &l)
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

// wait until at most VM of this wave's vector-memory operations are outstanding (they complete in issue order) and its LDS operations
// have returned, then the workgroup barrier
#define CR_BARRIER(VM)                                                        \
    do {                                                                      \
        asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
    } while (0)
#define CR_BARRIER_LDS()                                                      \
    do {                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
        __builtin_amdgcn_s_barrier();                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
    } while (0)

template <int N2, bool HAS_X2, int NTERMS>
__global__ __launch_bounds__(256, 2) void hvn_conv_chain_x3r(const ChainArgs p)
{
    constexpr int BM = CR_BM, CB = CR_CB;
    constexpr int KT = HAS_X2 ? 4 : 2;                // k-steps of 32 of the first GEMM: K1 = 64 (+ K1b = 64)
    constexpr int NS = 2 * KT;                        // its 16-deep slices
    constexpr int EP_BYTES = BM * CB * 4;             // epilogue / GEMM2-A tile [128][32] floats
    constexpr int W1_KSTEP = 3 * CB * 64;             // one k-step of a W1 chunk: [plane 3][row 32][64 B]
    constexpr int W1_BYTES = KT * W1_KSTEP;           // 12 | 24 KB
    constexpr int W2_BYTES = 3 * N2 * 64;             // one k-step of W1': [plane 3][row N2][64 B] = 12 | 24 KB
    constexpr int W1_OFF = EP_BYTES, W2_OFF = W1_OFF + W1_BYTES;
    constexpr int D1 = W1_BYTES / 4096, D2 = W2_BYTES / 4096;   // LDS-DMA instructions (1 KiB) per wave and chunk: 3 | 6
    constexpr int IPP2 = N2 / 16;                     // DMA instructions per W1' plane
    constexpr int WAVES_M2 = N2 == 128 ? 2 : 4, WAVES_N2 = 4 / WAVES_M2;
    constexpr int WM2 = BM / WAVES_M2, WN2 = N2 / WAVES_N2;
    constexpr int TM2 = WM2 / 32, TN2 = WN2 / 32;     // 1 x 2 | 2 x 2
    constexpr int NIT = BM / 32;                      // epilogue passes: 32 rows x 8 pieces of 16 B per pass
    constexpr bool RES = !HAS_X2;                     // a fused shortcut takes the residual's place (validated by the launcher)
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NTERMS == 9 || NTERMS == 6, "nine exact partial products, or the six that carry > 2^-24 of the product");
    static_assert(NIT == 4, "the counted waits below assume 4 y stores and 4 residual loads per thread and chunk");
    extern __shared__ __attribute__((aligned(16))) unsigned char cs[];
    float *ep = (float *)cs;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned M = (unsigned)p.M;
    const unsigned m0 = blockIdx.x * (unsigned)BM;
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const unsigned n_blk = m0 / HoWo;
    const int NC = p.C / CB;
    const int KT2 = p.C / 32;                         // k-steps of the second GEMM (row length of W1')
    const bool has_res = p.res != nullptr;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (long)n_blk * p.xsn), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x2 =
        __builtin_amdgcn_make_buffer_rsrc((void *)(HAS_X2 ? p.x2 + (long)n_blk * p.x2sn : p.x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc((void *)p.w2, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void *)(p.y + (long)n_blk * p.ysn), 0, 0x7fffffff, 0x00020000);
    // Optional operands are loaded UNCONDITIONALLY through descriptors of zero records when absent (every load out of range: zeros, no
    // memory traffic): the number of vector-memory operations per chunk is then the same on every path, which the counted waits
    // below -- mine and the compiler's -- depend on.
    const __amdgpu_buffer_rsrc_t rsrc_r =
        __builtin_amdgcn_make_buffer_rsrc((void *)(has_res ? p.res + (long)n_blk * p.rsn : p.x), 0, has_res ? 0x7fffffff : 0, 0x00020000);
    const bool has_post = p.post_s != nullptr, has_pre = p.pre_s != nullptr;
    const __amdgpu_buffer_rsrc_t rsrc_qs = __builtin_amdgcn_make_buffer_rsrc((void *)(has_post ? p.post_s : p.x), 0, has_post ? 0x7fffffff : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_qb = __builtin_amdgcn_make_buffer_rsrc((void *)(has_post ? p.post_b : p.x), 0, has_post ? 0x7fffffff : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_ps = __builtin_amdgcn_make_buffer_rsrc((void *)(has_pre ? p.pre_s : p.x), 0, has_pre ? 0x7fffffff : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_pb = __builtin_amdgcn_make_buffer_rsrc((void *)(has_pre ? p.pre_b : p.x), 0, has_pre ? 0x7fffffff : 0, 0x00020000);

    // ---- weight chunks by LDS-DMA.  Instruction t = wave D + jj fills the 1 KiB of LDS slots t 64 + lane (16 B each), lane-linear; the
    //      swizzle is applied to the SOURCE address: slot (plane, row, physical piece) receives the row's logical piece phys ^ ((row >> 2) & 3).
    //      W1 chunk: [k-step][plane 3][row 32][4 pieces] = 6 instructions per k-step; W1' k-step: [plane 3][row N2][4 pieces].
    unsigned w1_voff[D1], w2_voff[D2];
#pragma unroll
    for (int jj = 0; jj < D1; ++jj) {
        const int t = wave * D1 + jj;
        const int kt = t / 6, u = t - kt * 6;
        const int plane = u >> 1;
        const int row = (u & 1) * 16 + (lane >> 2);
        const int piece = (lane & 3) ^ ((row >> 2) & 3);
        w1_voff[jj] = (unsigned)(row * KT * 192 + kt * 192 + plane * 64 + piece * 16);
    }
#pragma unroll
    for (int jj = 0; jj < D2; ++jj) {
        const int t = wave * D2 + jj;
        const int plane = t / IPP2;
        const int row = (t - plane * IPP2) * 16 + (lane >> 2);
        const int piece = (lane & 3) ^ ((row >> 2) & 3);
        w2_voff[jj] = (unsigned)((long)row * KT2 * 192 + plane * 64 + piece * 16);
    }
    auto issue_w = [&](int c) {        // W1 chunk c (all k-steps) and k-step c of W1' (into stage c & 1)
        const int s1 = __builtin_amdgcn_readfirstlane(c * (CB * KT * 192));
        const int s2 = __builtin_amdgcn_readfirstlane(c * 192);
#pragma unroll
        for (int jj = 0; jj < D1; ++jj) cr_dma16(rsrc_w1, (cr_lds_ptr_t)(cs + W1_OFF + (wave * D1 + jj) * 1024), w1_voff[jj], s1);
#pragma unroll
        for (int jj = 0; jj < D2; ++jj)
            cr_dma16(rsrc_w2, (cr_lds_ptr_t)(cs + W2_OFF + (c & 1) * W2_BYTES + (wave * D2 + jj) * 1024), w2_voff[jj], s2);
        // nothing that is issued later in program order may be moved ahead of the DMAs: the counted waits below count what is YOUNGER
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    issue_w(0);

    // ---- conv3's input rows of this wave (32 pixels), straight into the MFMA A-fragment layout: lane (l31, lh) holds row l31's
    //      k = 16 s + 8 lh .. + 7 of every 16-deep slice s; split once -----------------------------------------------------------
    bf16x8 fa[NS][3];
    f32x4 raw[NS][2];
    {
        const unsigned m = m0 + wave * 32 + l31;
        const bool ok = m < M;
        const unsigned mm = ok ? m : m0;
        const unsigned n = mm / HoWo;
        const unsigned rem = mm - n * HoWo;
        const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
        const unsigned ax = ok ? (unsigned)(((long)(n - n_blk) * p.xsn + (long)oy * p.xsy + (long)ox * p.xsx + lh * 8) * 4) : OOB;
        const unsigned ax2 = (ok && HAS_X2) ? (unsigned)(((long)(n - n_blk) * p.x2sn + (long)(oy * p.stride2) * p.x2sy + (long)(ox * p.stride2) * p.x2sx + lh * 8) * 4) : OOB;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool second = HAS_X2 && s >= 4;
            const int soff = (second ? s - 4 : s) * 64;
            raw[s][0] = cr_load(second ? rsrc_x2 : rsrc_x, second ? ax2 : ax, soff);
            raw[s][1] = cr_load(second ? rsrc_x2 : rsrc_x, (second ? ax2 : ax) + 16, soff);
        }
    }

    // ---- epilogue coordinates: thread = (16-byte column piece tid & 7, rows (tid >> 3) + 32 it); tile slot of (row, piece) =
    //      piece ^ ((row >> 1) & 7) -- the key does not depend on `it` -----------------------------------------------------------
    const int ecol = (tid & 7) * 4;
    const int erow0 = tid >> 3;
    const int e_off = erow0 * CB + (((tid & 7) ^ ((erow0 >> 1) & 7)) << 2);     // floats; + 32 it rows
    unsigned y_voff[NIT];      // the residual view has the output's strides (validated by the launcher): same offsets, other base
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const unsigned m = m0 + erow0 + 32 * it;
        const bool ok = m < M;
        const unsigned mm = ok ? m : m0;
        const unsigned n = mm / HoWo;
        const unsigned rem = mm - n * HoWo;
        const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
        y_voff[it] = ok ? (unsigned)(((long)(n - n_blk) * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + ecol) * 4) : OOB;
    }
    const float post_lo = has_post ? 0.f : -__builtin_inff();
    const float pre_lo = has_pre ? 0.f : -__builtin_inff();

    f32x4 rres[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) rres[it] = RES ? cr_load(rsrc_r, y_voff[it], 0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);         // (the first residual chunk is requested before the wait for the input rows)
#pragma unroll
    for (int s = 0; s < NS; ++s) cr_split(raw[s][0], raw[s][1], fa[s][0], fa[s][1], fa[s][2]);

    f32x16 acc2[TM2][TN2];
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    const int akey = (l31 >> 1) & 7, bkey = (l31 >> 2) & 3;
    const int wm2 = wave / WAVES_N2, wn2 = wave % WAVES_N2;

    // the partial products of one 16-deep slice with a-plane + b-plane = s, for one accumulator, smallest first (hvn_conv_igemm_x3's order)
    auto mac = [&](f32x16 &acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3], int s) {
#pragma unroll
        for (int pa = 2; pa >= 0; --pa) {
            const int pb = s - pa;
            if (pb < 0 || pb > 2) continue;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[pb], acc, 0, 0, 0);
        }
    };

    // the wave's own DMAs of chunk 0 are older than everything but the 4 residual loads, which stay in flight
    if constexpr (RES)
        CR_BARRIER(4);
    else
        CR_BARRIER(0);

    auto chunk = [&](const int c, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        // the chunk's per-channel vectors: loaded HERE (ahead of the next chunk's DMAs, which are issued behind the barrier after GEMM 1) --
        // vector-memory operations complete in issue order, so waiting for a load issued after the DMAs would wait for the DMAs
        const f32x4 ones = {1.f, 1.f, 1.f, 1.f};
        const int csoff = __builtin_amdgcn_readfirstlane(c * (CB * 4));
        f32x4 qs = cr_load(rsrc_qs, ecol * 4, csoff), qb = cr_load(rsrc_qb, ecol * 4, csoff);
        f32x4 ps = cr_load(rsrc_ps, ecol * 4, csoff), pb = cr_load(rsrc_pb, ecol * 4, csoff);
        __builtin_amdgcn_sched_barrier(0);     // (issued here; their first use is behind the next barrier -- hipcc otherwise waits for them, and with
                                               //  them for the residual loads ahead of them, in the middle of GEMM 1)
        // ---- GEMM 1: this wave's 32 pixels x the chunk's 32 channels; A from registers, W1 planes from LDS --------------------------
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const unsigned char *b = cs + W1_OFF + (s >> 1) * W1_KSTEP + l31 * 64 + (((2 * (s & 1) + lh) ^ bkey) << 4);
            bf16x8 fb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[pl] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(b + pl * (CB * 64)));
#pragma unroll
            for (int t = 4; t >= 0; --t) {
                if (NTERMS == 6 && t > 2) continue;
                mac(acc1, fa[s], fb, t);
            }
        }
        // accumulators -> tile (free since the barrier behind the previous chunk's GEMM 2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            ep[row * CB + (((l31 >> 2) ^ ((row >> 1) & 7)) << 2) + (l31 & 3)] = acc1[r];
        }
        CR_BARRIER_LDS();              // tile visible; every wave is done with this chunk's W1 planes (and with W1' stage (c + 1) & 1)
        if constexpr (!LAST) issue_w(c + 1);
        qs = has_post ? qs : ones;
        ps = has_pre ? ps : ones;
        // ---- epilogue 1: + residual, block-closing BN-ReLU, store y, next unit's pre-activation back into the tile ------------------
        {
            f32x4 vout[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float *e = ep + e_off + 32 * it * CB;
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
#pragma unroll
            for (int it = 0; it < NIT; ++it) cr_store(vout[it], rsrc_y, y_voff[it], csoff);
            if constexpr (!LAST && RES) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) rres[it] = cr_load(rsrc_r, y_voff[it], csoff + CB * 4);
            }
        }
        CR_BARRIER_LDS();              // the activated tile is visible
        // ---- GEMM 2: k-step c of t1' += act(y chunk) W1'^T; A = the tile (fp32, split at the fragment read), B = W1' stage c & 1 -----
        {
            const float *a = ep + (wm2 * WM2 + l31) * CB;
            const unsigned char *b = cs + W2_OFF + (c & 1) * W2_BYTES + (wn2 * WN2 + l31) * 64;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bf16x8 fa2[TM2][3], fb2[TN2][3];
#pragma unroll
                for (int i = 0; i < TM2; ++i) {
                    const f32x4 v0 = *(const f32x4 *)(a + i * 32 * CB + (((4 * q + 2 * lh) ^ akey) << 2));
                    const f32x4 v1 = *(const f32x4 *)(a + i * 32 * CB + (((4 * q + 2 * lh + 1) ^ akey) << 2));
                    cr_split(v0, v1, fa2[i][0], fa2[i][1], fa2[i][2]);
                }
#pragma unroll
                for (int j = 0; j < TN2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fb2[j][pl] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(b + pl * (N2 * 64) + j * 32 * 64 + (((2 * q + lh) ^ bkey) << 4)));
#pragma unroll
                for (int t = 4; t >= 0; --t) {
                    if (NTERMS == 6 && t > 2) continue;
#pragma unroll
                    for (int i = 0; i < TM2; ++i)
#pragma unroll
                        for (int j = 0; j < TN2; ++j) mac(acc2[i][j], fa2[i], fb2[j], t);
                }
            }
        }
        // every wave is done with the tile and with W1' stage c & 1; the next chunk's weights (issued behind the barrier after GEMM 1)
        // have landed: younger than them are this chunk's 4 y stores and, if any, the 4 residual loads of the next chunk
        if constexpr (LAST)
            CR_BARRIER_LDS();
        else if constexpr (RES)
            CR_BARRIER(8);
        else
            CR_BARRIER(4);
    };
    for (int c = 0; c + 1 < NC; ++c) chunk(c, std::false_type{});
    chunk(NC - 1, std::true_type{});

    // ---- epilogue 2: t1' = relu(acc2 + b2), 32 output channels at a time through the tile ----------------------------------------
    unsigned y2_voff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const unsigned m = m0 + erow0 + 32 * it;
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
    for (int h = 0; h < N2 / 32; ++h) {
#pragma unroll
        for (int j = 0; j < TN2; ++j) {
            if ((wn2 * WN2 + j * 32) / 32 == h) {       // this wave's 32-column tile j is the group being written out
#pragma unroll
                for (int i = 0; i < TM2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm2 * WM2 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        ep[row * CB + (((l31 >> 2) ^ ((row >> 1) & 7)) << 2) + (l31 & 3)] = acc2[i][j][r];
                    }
            }
        }
        CR_BARRIER_LDS();
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (p.bias2) bias = *(const f32x4 *)(p.bias2 + h * 32 + ecol);
        f32x4 vout[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f32x4 v = *(const f32x4 *)(ep + e_off + 32 * it * CB);
            v.x = fmaxf(v.x + bias.x, relu_lo);
            v.y = fmaxf(v.y + bias.y, relu_lo);
            v.z = fmaxf(v.z + bias.z, relu_lo);
            v.w = fmaxf(v.w + bias.w, relu_lo);
            vout[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) cr_store(vout[it], rsrc_y2, y2_voff[it], h * 128);
        if (h + 1 < N2 / 32) CR_BARRIER_LDS();
    }
}

template <int N2, bool HAS_X2, int NTERMS>
static int launch_chain_x3r(const ChainArgs &a, hipStream_t stream)
{
    constexpr size_t lds = (size_t)CR_BM * CR_CB * 4 + (size_t)(HAS_X2 ? 4 : 2) * 3 * CR_CB * 64 + (size_t)2 * 3 * N2 * 64;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_conv_chain_x3r<N2, HAS_X2, NTERMS>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long grid = (a.M + CR_BM - 1) / CR_BM;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Does the op have this form?  conv3 with 64 input channels (+ 64 of a fused shortcut, then without a residual view and with cout2 = 64):
// the input tile's planes must fit the register file next to the accumulators.
int hvn_chain_x3r_supported(const ChainArgs &a)
{
    if (!hvn_chain_supported(a.C, a.N2) || a.K1 != 64) return 0;
    if (a.x2 && (a.K1b != 64 || a.res || a.N2 != 64)) return 0;
    if (!a.x2 && a.K1b) return 0;
    return 1;
}

// ChainArgs as hvn_launch_conv_chain_x3 (same operands, same packings, same bits).
int hvn_launch_conv_chain_x3r(const ChainArgs &a, int terms, hipStream_t stream)
{
    if (!hvn_chain_x3r_supported(a)) return -1;
    if (a.M <= 0 || a.M >= (1L << 31) - 256) return -1;
    const long px = (long)a.Ho * a.Wo;
    if (px <= 0) return -1;
    const long ns = 128 / px + 2;
    const long spans[5] = {ns * a.xsn, a.x2 ? ns * a.x2sn : 0, ns * a.ysn, a.res ? ns * a.rsn : 0, ns * a.y2sn};
    for (long s : spans)
        if (s < 0 || s * 4 >= (1L << 31)) return -1;
    if ((long)(a.C + 64) * ((a.K1 + a.K1b) / 32) * 192 >= (1L << 31) || (long)(a.N2 + 64) * (a.C / 32) * 192 >= (1L << 31)) return -1;
    const bool nine = terms != 6;
    if (a.x2) return nine ? launch_chain_x3r<64, true, 9>(a, stream) : launch_chain_x3r<64, true, 6>(a, stream);
    if (a.N2 == 64) return nine ? launch_chain_x3r<64, false, 9>(a, stream) : launch_chain_x3r<64, false, 6>(a, stream);
    return nine ? launch_chain_x3r<128, false, 9>(a, stream) : launch_chain_x3r<128, false, 6>(a, stream);
}
