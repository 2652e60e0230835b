// hvn_conv.hip -- fused implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
// Stands behind every nn.Conv2d (+ the BatchNorm/ReLU/residual/concat/crop ops around
// it) of /root/reference/models/hovernet/net_desc.py:101-145 except conv0 and the 1x1
// logit heads (hvn_net_ops.hip).
//
// GEMM view:  D[m][co] = sum_k A[m][k] * Wt[co][k],   m = (n, oy, ox) output pixel,
//             k = (32-channel slab, tap, ci).  Activations are channels-last, so one
//             k-chunk of 32 is 128 contiguous bytes of A; weights are pre-packed
//             [cout_pad][cin/32][taps][32] so the same holds for B.  All taps of one slab
//             are walked before the next slab: the shifted windows of a 5x5 / 3x3 filter
//             re-read the same ~50 KB of input from L1/L2 instead of HBM.
// Tile:       128 pixels x {128,64,32} output channels per 256-thread workgroup,
//             BK = 32, double-buffered LDS.  The global loads of k-step t+1 are issued
//             RAW before the MFMAs of step t and only touched (prologue BN+ReLU, zero
//             padding select, ds_write) after them, so HBM/L2 latency hides under
//             >=1024 matrix-pipe cycles per wave.
// MFMA:       v_mfma_f32_32x32x2_f32, exact fp32 (== an fmaf chain).  Lane l feeds
//             A[i=l&31][k=l>>5] / B[k=l>>5][j=l&31]; the k-ORDER inside a chunk of 8
//             is permuted (lanes <32 take k 0..3, lanes >=32 take k 4..7) so that each
//             lane reads its 4 operands with ONE ds_read_b128 -- a sum is a sum.
// LDS:        rows of 32 floats with an XOR chunk swizzle (see LDS_LD below): the 16-lane
//             groups of ds_read_b128 and ds_write_b128 hit disjoint banks without padding
//             (MI355X_MICROARCH section LDS), and three 128x64 workgroups fit one CU.
// Epilogue:   accumulators are transposed through LDS (the k-loop is done with it) so
//             that every global access is 16 B per lane and 512 B contiguous per row:
//             bias / ReLU / residual add / block-closing BN-ReLU are applied on float4s.
// Grid:       1-D, XCD-aware: workgroups that share an A tile (same pixels, different
//             cout tile) get the same blockIdx % 8, i.e. the same XCD L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include <type_traits>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: HIP's float4 struct copies lower to memcpy -> scratch

#define BK 32
// LDS row layout of the staged A / B tiles (rows of BK = 32 floats = eight 16-byte chunks):
//   HVN_SWZ=1  (default) unpadded rows of 32 floats, chunk c of row r stored at chunk c ^ ((r >> 1) & 7): the 16 lanes of one
//              ds_read_b128 phase (16 consecutive rows, one logical chunk) cover all 64 banks -- rows 2k / 2k+1 share a chunk slot
//              but lie in opposite bank halves.  The staging buffers are 1/9 smaller than with padding: a 128x64 tile needs 48 KB,
//              so THREE workgroups fit a CU's 160 KB (128x32: four); measured in profiles/r02_experiments.md section 8.
//   HVN_SWZ=0  rows padded to 36 floats (16-byte skew per row, the round-1 layout; 54 KB per 128x64 tile = two per CU).  Kept as
//              the A/B build `libhvn_hip_pad.so` (lib.VARIANTS).
#ifndef HVN_SWZ
#define HVN_SWZ 1
#endif
// Experiment builds (lib.VARIANTS, default 0: the default library is byte-identical without them; NOT yet measured -- prepared for the next GPU session):
//   HVN_EPI_LINEAR=1  branch-free epilogue addressing for row-contiguous output / residual views (offset = base[sample] + m * pixel stride)
//   HVN_NT=1          non-temporal hints on the epilogue's residual loads and output stores (read once / written once per launch)
#ifndef HVN_EPI_LINEAR
#define HVN_EPI_LINEAR 0
#endif
#ifndef HVN_NT
#define HVN_NT 0
#endif
//   HVN_TRACE_FINE=1  (with HVN_CONV_TRACE) 8 instead of 4 words per workgroup: + the end of each epilogue phase (accumulators in LDS and
//                     barrier passed, residual loads returned, values finished, stores issued); the phase waits it inserts perturb the timing
//                     a little -- a diagnosis build (tools/conv_trace.py --fine)
#ifndef HVN_TRACE_FINE
#define HVN_TRACE_FINE 0
#endif
#define HVN_TRACE_WORDS (HVN_TRACE_FINE ? 8 : 4)
#if HVN_SWZ
#define LDS_LD 32
#else
#define LDS_LD 36  // padded row length in floats
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PADDED, int ABL = 0, bool GROUPED = false, bool HAS_PRE = true, bool HAS_X2 = false>
__global__ __launch_bounds__(256, (HVN_SWZ && BN <= 64 && BM <= 128) ? 3 : 2) void hvn_conv_igemm_f32(ConvArgs p)
{
    // EXPERIMENT (off by default, HVN_STAGGER=1|2|3): two workgroups share a CU, i.e. two waves share each SIMD's matrix pipe; giving
    // one of them a raised issue priority (told apart by LDS base or wave slot) or a delayed start was meant to keep one
    // workgroup's epilogue under the other's k-loop.  Measured: no effect (profiles/r02_experiments.md section 7).
    if (p.stagger) {
        // the two workgroups of a CU are told apart by where their LDS allocation starts (HW_REG_LDS_ALLOC[7:0] = LDS_BASE)
        const unsigned lds_base = __builtin_amdgcn_s_getreg(6 | (0 << 6) | ((8 - 1) << 11));
        if (p.stagger == 1 && lds_base != 0u) __builtin_amdgcn_s_setprio(3);
        if (p.stagger == 2) {               // variant: by the wave slot on the SIMD (HW_REG_HW_ID[3:0])
            const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((4 - 1) << 11));
            if (slot & 1u) __builtin_amdgcn_s_setprio(3);
        }
        if (p.stagger == 3 && lds_base != 0u) __builtin_amdgcn_s_sleep(127);   // variant: the second workgroup starts ~8k cycles late (no priority)
    }
    unsigned long long t_start = 0, t_kend = 0;
    if (p.dbg) t_start = __builtin_readcyclecounter();
    if (p.nbatch > 1) {  // batched launch: one of nbatch independent problems per blockIdx.y
        p.x += (long)blockIdx.y * p.xb;
        p.w += (long)blockIdx.y * p.wb;
        p.y += (long)blockIdx.y * p.yb;
    }
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;  // staging passes (32 rows of 8 float4 per pass)
    constexpr int EP_LD = BN + 4;              // epilogue tile row length (floats)
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");
    static_assert(!GROUPED || (BN == 32 && WM == 32 && WN == 32), "grouped mode: 4 groups x 8 output channels per 32-wide tile");
    // dynamic LDS = max(staging buffers, epilogue tile): see launch_conv
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
    // (multi-tap launches: neighbouring pixel tiles share input rows, so XCD k takes the k-th CONTIGUOUS eighth of them and finds the
    //  halo in its own L2; in round-robin order every XCD fetched it from HBM for itself -- measured 2.1 - 3.3 x the compulsory reads)
    const int m_tile = hvn_m_tile(xcd, seq / NT, (int)(gridDim.x / (8u * (unsigned)NT)), p.KH * p.KW > 1);
    if (m_tile >= (int)p.m_tiles) return;
    const unsigned m0 = (unsigned)m_tile * BM;
    const int n0 = n_tile * BN;
    const unsigned M = (unsigned)p.M;

    // ---- per-thread staging coordinates --------------------------------------------
    // Addressing is split into a wave-UNIFORM 64-bit base that moves with the k-step (scalar ALU)
    // and a per-thread 32-bit byte offset that never changes (global_load saddr + voffset form):
    // the k-loop then carries no vector address arithmetic at all.
    const int srow = tid >> 3;      // 0..31
    const int scol = (tid & 7) * 4; // float offset inside the 32-wide k chunk
    const int lcol = HVN_SWZ ? (((tid & 7) ^ ((srow >> 1) & 7)) * 4) : scol;  // ... and where that chunk lives in the LDS row
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const unsigned n_blk = m0 / HoWo;                                        // sample of the tile's first row
    const long padoff = (long)p.pad_t * p.xsy + (long)p.pad_l * p.xsx;       // keeps every thread offset >= 0
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
        a_voff[j] = ok ? (unsigned)(((long)(n - n_blk) * p.xsn + (long)(oy * p.stride) * p.xsy + (long)(ox * p.stride) * p.xsx + scol) * 4)
                       : OOB;  // rows past the end of the batch load zeros (buffer range check)
    }
    const float *xblk = p.x + (long)n_blk * p.xsn - padoff;
    // Buffer descriptors: address = base + soffset (SGPR, moves with the k-step) + voffset (VGPR, loop-invariant),
    // so a load carries NO vector address arithmetic; voffset >= num_records makes the hardware return 0, which is
    // how out-of-image taps (zero padding) and tail rows are produced without a select.
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)xblk, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, 0x7fffffff, 0x00020000);
    // optional second 1x1 input (fused shortcut): its channels extend the reduction after the first input's
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
    const long Ktot = (long)p.KH * p.KW * p.Cin + (HAS_X2 ? p.Cin2 : 0);
    unsigned w_voff[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) w_voff[j] = (unsigned)(((long)(n0 + srow + 32 * j) * Ktot + scol) * 4);

    const int kchunks = p.Cin / BK;
    const int KT1 = p.KH * p.KW * kchunks;
    const int KT = KT1 + (HAS_X2 ? p.Cin2 / BK : 0);
    const bool has_pre = HAS_PRE && p.pre_s != nullptr;
    // branch-free prologue: scale 1 / shift 0 / clamp -inf when there is none
    const float pre_lo = has_pre ? 0.f : -__builtin_inff();
    const float *pre_s = has_pre ? p.pre_s : p.w;  // any valid address; value unused when !has_pre
    const float *pre_b = has_pre ? p.pre_b : p.w;
    const int pre_step = has_pre ? BK : 0;

    // two register stages: the loads of k-step t+2 are issued while step t computes and are first
    // touched in the tail of step t+1 (>= one full step of matrix work hides HBM / Infinity-Cache latency)
    struct Stage {
        f32x4 ra[PA], rb[PB], rps, rpb;
    };
    Stage s0, s1;
    int ld_r = 0, ld_s = 0, ld_c = 0;  // tap row / col / channel slab of the NEXT load

    // issue the raw loads of one k-step (nothing here waits on memory)
    auto load_global = [&](Stage &st, int kt) {
        // uniform (SALU).  Channels-last: slab ld_c starts at channel 32 ld_c.  Blocked experiment: block (ld_c >> s) + slab (ld_c & mask) inside it
        const unsigned cs = p.blk_shift ? (unsigned)p.blk_shift - 5u : 31u;
        int a_soff = (int)(((long)ld_r * p.xsy + (long)ld_s * p.xsx + (long)((unsigned)ld_c >> cs) * p.xsb + (long)((unsigned)ld_c & ((1u << cs) - 1u)) * BK) * 4);
        const int w_soff = kt * (BK * 4);
        const bool second = HAS_X2 && kt >= KT1;                                             // uniform
        if constexpr (HAS_X2) a_soff = second ? (kt - KT1) * (BK * 4) : a_soff;
        if constexpr (HAS_PRE) {
            st.rps = *(const f32x4 *)(pre_s + ld_c * pre_step + scol);
            st.rpb = *(const f32x4 *)(pre_b + ld_c * pre_step + scol);
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
                st.ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? rsrc_a2 : rsrc_a, vo, a_soff, 0));
            } else
                st.ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, a_soff, 0));
        }
#pragma unroll
        for (int j = 0; j < PB; ++j)
            st.rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[j], w_soff, 0));
        // advance: tap column, tap row, then the next 32-channel slab
        if (++ld_s == p.KW) {
            ld_s = 0;
            if (++ld_r == p.KH) {
                ld_r = 0;
                ++ld_c;
            }
        }
    };
    // first touch of the loaded registers: pre-activation BN+ReLU, zero padding, park in LDS
    auto store_lds = [&](Stage &st, int buf) {
        float *a = As + buf * BM * LDS_LD;
        float *b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            f32x4 v = st.ra[j];
            if constexpr (HAS_PRE) {
                // pre-activation BN + ReLU (scale 1 / shift 0 / clamp -inf when the op has none).  Zero padding
                // would have to be re-applied after it; the host rejects prologue + padding (never needed here).
                f32x4 ps = st.rps, pb = st.rpb;
                if (!has_pre) {
                    ps = (f32x4){1.f, 1.f, 1.f, 1.f};
                    pb = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                v.x = fmaxf(fmaf(v.x, ps.x, pb.x), pre_lo);
                v.y = fmaxf(fmaf(v.y, ps.y, pb.y), pre_lo);
                v.z = fmaxf(fmaf(v.z, ps.z, pb.z), pre_lo);
                v.w = fmaxf(fmaf(v.w, ps.w, pb.w), pre_lo);
            }
            *(f32x4 *)(a + (srow + 32 * j) * LDS_LD + lcol) = v;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) *(f32x4 *)(b + (srow + 32 * j) * LDS_LD + lcol) = st.rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // GROUPED (block-diagonal weights, 8 output channels per 32-channel slab): 16x16x4 MFMA tiles, two pixel
    // tiles x two 16-channel column tiles per wave; a slab only touches the column tile that holds its group
    f32x4 gacc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) gacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int taps = p.KH * p.KW;

    // MFMAs of the k-sub-chunks [Q0, Q1) (8 reduction indices each) of LDS buffer `cur`
    auto compute_grouped = [&](int cur, int kt, auto q0c, auto q1c) {
        constexpr int Q0 = decltype(q0c)::value, Q1 = decltype(q1c)::value;  // in units of 8 k
        const int l15 = lane & 15, l4 = lane >> 4;
        const int grp = kt / taps;                 // slab == group (32 input channels per group)
        const int nt = grp >> 1;                   // column tile holding this group's 8 output channels
        const float *a = As + cur * BM * LDS_LD + (wm * WM + l15) * LDS_LD;
        const float *b = Bs + cur * BN * LDS_LD + (nt * 16 + l15) * LDS_LD;
        const int key = (l15 >> 1) & 7;             // rows +16 / +nt*16 leave bits 1..3 of the row alone
#pragma unroll
        for (int c = Q0 / 2; c < (Q1 + 1) / 2; ++c) {  // chunks of 16 k (4 lane groups x 4)
            const int off = HVN_SWZ ? (((l4 + 4 * c) ^ key) * 4) : (4 * l4 + c * 16);
            const f32x4 fb = *(const f32x4 *)(b + off);
            const f32x4 fa0 = *(const f32x4 *)(a + off);
            const f32x4 fa1 = *(const f32x4 *)(a + 16 * LDS_LD + off);
            if (nt == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gacc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[e], fb[e], gacc[0][0], 0, 0, 0);
                    gacc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[e], fb[e], gacc[1][0], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gacc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[e], fb[e], gacc[0][1], 0, 0, 0);
                    gacc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[e], fb[e], gacc[1][1], 0, 0, 0);
                }
            }
        }
    };
    auto compute = [&](int cur, auto q0c, auto q1c) {
        constexpr int Q0 = decltype(q0c)::value, Q1 = decltype(q1c)::value;
        const float *a = As + cur * BM * LDS_LD + (wm * WM + l31) * LDS_LD;
        const float *b = Bs + cur * BN * LDS_LD + (wn * WN + l31) * LDS_LD;
        const int key = (l31 >> 1) & 7;             // the +32-row steps and the wave's row base leave bits 1..3 of the row alone
#pragma unroll
        for (int q = Q0; q < Q1; ++q) {
            f32x4 fa[TM], fb[TN];
            const int off = HVN_SWZ ? (((2 * q + lh) ^ key) * 4) : (q * 8 + 4 * lh);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const f32x4 *)(a + i * 32 * LDS_LD + off);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const f32x4 *)(b + j * 32 * LDS_LD + off);
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
    auto mma = [&](int kt, auto q0c, auto q1c) {
        if constexpr (GROUPED) compute_grouped(kt & 1, kt, q0c, q1c);
        else compute(kt & 1, q0c, q1c);
    };
    using std::integral_constant;
    constexpr int NQ = BK / 8;
    constexpr int QT = (16 / (TM * TN * 4)) > 0 ? ((16 / (TM * TN * 4)) < NQ ? (16 / (TM * TN * 4)) : NQ) : 1;  // tail sub-chunks (>= 16 MFMAs)
    constexpr int TAIL_MFMA = QT * TM * TN * 4;

    // one k-step, three scheduling segments (sched_barrier pins their order, sched_group_barrier the
    // interleave inside):
    //   (1) first MFMAs of the step, each followed by one of the global loads for step kt+2
    //       (a VMEM issue stalls its wave for tens of cycles: hidden in the 64-cycle MFMA shadow)
    //   (2) middle MFMAs carrying the first touch of the stage loaded one step ago: prologue BN+ReLU,
    //       padding select, ds_write into the other LDS buffer
    //   (3) remaining MFMAs; by the barrier every wave's LDS writes are long retired
    auto step = [&](Stage &ld, Stage &stg, int kt, auto do_load) {
        constexpr int PER_Q = TM * TN * 4;                       // MFMAs per k-sub-chunk of 8
        constexpr int Q1 = (NQ >= 4 && PER_Q >= 8) ? 1 : NQ / 2;  // sub-chunks in segment 1
        constexpr int Q2 = (NQ >= 4 && PER_Q >= 8) ? 3 : NQ;      // end of segment 2
        if constexpr (ABL == 3) {
            f32x4 fa0 = stg.ra[0], fb0 = stg.rb[0];  // pure MFMA: fragments held in registers
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.x, fb0.x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.y, fb0.y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.z, fb0.z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.w, fb0.w, acc[i][j], 0, 0, 0);
                    }
            return;
        }
        // ---- segment 1 ----
        mma(kt, integral_constant<int, 0>{}, integral_constant<int, Q1>{});
        if constexpr (decltype(do_load)::value && ABL < 1) load_global(ld, kt + 2);
#pragma unroll
        for (int g = 0; g < Q1 * PER_Q; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
            __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);  // VALU | SALU (address of the next load)
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- segment 2 ----
        mma(kt, integral_constant<int, Q1>{}, integral_constant<int, Q2>{});
        if constexpr (ABL < 2) {
            store_lds(stg, (kt + 1) & 1);
#pragma unroll
            for (int g = 0; g < (Q2 - Q1) * PER_Q; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- segment 3 ----
        if constexpr (Q2 < NQ) mma(kt, integral_constant<int, Q2>{}, integral_constant<int, NQ>{});
        if constexpr (ABL < 2) __syncthreads();
    };
    const std::true_type LOAD{};
    const std::false_type NOLOAD{};

    // data of k-step j travels in stage (j & 1)
    load_global(s0, 0);
    store_lds(s0, 0);
    if (KT > 1) load_global(s1, 1);
    __syncthreads();
    int kt = 0;
    for (; kt + 3 < KT; kt += 2) {  // steps kt (even) and kt+1 (odd), both with a load two ahead
        step(s0, s1, kt, LOAD);
        step(s1, s0, kt + 1, LOAD);
    }
    if (kt + 2 < KT) {              // one more loading step (kt even)
        step(s0, s1, kt, LOAD);
        ++kt;
        if (kt + 1 < KT) {          // step KT-2 (odd): stage of step KT-1 is s0
            step(s1, s0, kt, NOLOAD);
            ++kt;
        }
    } else if (kt + 1 < KT) {       // step KT-2 (even): stage of step KT-1 is s1
        step(s0, s1, kt, NOLOAD);
        ++kt;
    }
    mma(KT - 1, integral_constant<int, 0>{}, integral_constant<int, NQ>{});
    __syncthreads();
    if (p.dbg) t_kend = __builtin_readcyclecounter();

    // ---- epilogue ------------------------------------------------------------------
    // 1. accumulators -> LDS tile [BM][EP_LD] (32 consecutive floats per half-wave: conflict-free)
    float *ep = smem;
    if constexpr (GROUPED) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ep[(wm * WM + i * 16 + (lane >> 4) * 4 + r) * EP_LD + j * 16 + (lane & 15)] = gacc[i][j][r];
    } else
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
#if HVN_TRACE_FINE
    unsigned long long t_f[4] = {0, 0, 0, 0};
    if (p.dbg) t_f[0] = __builtin_readcyclecounter();
#endif
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
    // this thread's rows are m0 + erow0 + k*RPP: decode the first with divisions, walk the rest (n, oy, ox) incrementally
    // (32 integer divisions per thread and tile were ~10 % of a short-K tile's instruction stream)
#if HVN_EPI_LINEAR
    // row-contiguous views: offset(n, oy, ox) = n * sn + (m - n * HoWo) * sx; a 128-row tile touches at most two samples (HoWo >= BM)
    const bool lin = p.ysy == (long)p.Wo * p.ysx && (!has_res || p.rsy == (long)p.Wo * p.rsx) && HoWo >= (unsigned)BM && !p.ysb && !p.rsb;   // uniform
    const unsigned lin_bound = (n_blk + 1u) * HoWo;                       // first pixel row of the next sample
    const long ybase0 = (long)n_blk * p.ysn - (long)n_blk * HoWo * p.ysx + co, ybase1 = ybase0 + p.ysn - (long)HoWo * p.ysx;
    const long rbase0 = (long)n_blk * p.rsn - (long)n_blk * HoWo * p.rsx + co, rbase1 = rbase0 + p.rsn - (long)HoWo * p.rsx;
#endif
#if HVN_EPI_LINEAR
    unsigned e_n = 0, e_oy = 0, e_ox = 0;
    if (!lin)
#else
    unsigned e_n, e_oy, e_ox;
#endif
    {
        const unsigned m = m0 + erow0;
        e_n = m / HoWo;
        const unsigned rem = m - e_n * HoWo;
        e_oy = rem / (unsigned)p.Wo;
        e_ox = rem - e_oy * (unsigned)p.Wo;
    }
    // Residual tile first: ALL of this thread's residual loads are issued before any store.  vmcnt retires loads and stores
    // in order, so a load issued after a store cannot be waited for without draining that store: the round-1 epilogue
    // (4 rounds of "4 loads, wait, 4 stores" -- with a dummy load per row when there is no residual) paid four loaded-memory
    // round trips per tile, 30-50 k cycles against a 14-40 k cycle k-loop on the K <= 512 layers (per-workgroup timelines,
    // profiles/r02_experiments.md).
    f32x4 rall[NIT];
#if HVN_EPI_LINEAR
    if (has_res && ABL != 4 && lin) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const unsigned m = m0 + erow0 + it * RPP;
            const long ro = (m < M && cok) ? (m >= lin_bound ? rbase1 : rbase0) + (long)m * p.rsx : 0;
#if HVN_NT
            rall[it] = __builtin_nontemporal_load((const f32x4 *)(p.res + ro));
#else
            rall[it] = *(const f32x4 *)(p.res + ro);
#endif
        }
    } else
#endif
    if (has_res && ABL != 4) {
        unsigned a_n = e_n, a_oy = e_oy, a_ox = e_ox;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const unsigned m = m0 + erow0 + it * RPP;
            const long ro = (m < M && cok) ? (long)a_n * p.rsn + (long)a_oy * p.rsy + (long)a_ox * p.rsx + (p.rsb ? (long)(co >> p.blk_shift) * p.rsb + (co & ((1 << p.blk_shift) - 1)) : (long)co) : 0;
#if HVN_NT
            rall[it] = __builtin_nontemporal_load((const f32x4 *)(p.res + ro));
#else
            rall[it] = *(const f32x4 *)(p.res + ro);
#endif
            a_ox += RPP;
            while (a_ox >= (unsigned)p.Wo) {
                a_ox -= (unsigned)p.Wo;
                ++a_oy;
            }
            while (a_oy >= (unsigned)p.Ho) {
                a_oy -= (unsigned)p.Ho;
                ++a_n;
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) rall[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // ... then every output value is finished in registers (the waits for the residual loads fall here, while no store is in
    // flight: with loads AND stores pending the compiler has to assume they retire out of order and waits for vmcnt(0), i.e.
    // for the previous store, before every use of a loaded value) ...
#if HVN_TRACE_FINE
    if (p.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the residual tile has arrived
        t_f[1] = __builtin_readcyclecounter();
    }
#endif
    f32x4 vout[NIT];
    long yoffs[NIT];
    bool oks[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const unsigned m = m0 + erow0 + it * RPP;
        oks[it] = m < M && cok;
        yoffs[it] = (long)e_n * p.ysn + (long)e_oy * p.ysy + (long)e_ox * p.ysx + (p.ysb ? (long)(co >> p.blk_shift) * p.ysb + (co & ((1 << p.blk_shift) - 1)) : (long)co);
#if HVN_EPI_LINEAR
        if (lin) yoffs[it] = (m >= lin_bound ? ybase1 : ybase0) + (long)m * p.ysx;
#endif
        if constexpr (ABL == 6) {   // experiment: the tile's 64 KB written as ONE contiguous block (wrong place, right amount)
            yoffs[it] = ((long)(m_tile * NT + n_tile) * BM + (erow0 + it * RPP)) * BN + ecol;
            oks[it] = oks[it] && yoffs[it] + 4 <= (long)M * p.Cout;
        }
#if HVN_EPI_LINEAR
        if (!lin)
#endif
        {
            e_ox += RPP;
            while (e_ox >= (unsigned)p.Wo) {
                e_ox -= (unsigned)p.Wo;
                ++e_oy;
            }
            while (e_oy >= (unsigned)p.Ho) {
                e_oy -= (unsigned)p.Ho;
                ++e_n;
            }
        }
        const int rr = erow0 + it * RPP;
        f32x4 v = *(const f32x4 *)(ep + rr * EP_LD + ecol);
        v.x = fmaxf(v.x + bias.x, relu_lo);
        v.y = fmaxf(v.y + bias.y, relu_lo);
        v.z = fmaxf(v.z + bias.z, relu_lo);
        v.w = fmaxf(v.w + bias.w, relu_lo);
        v += rall[it];
        v.x = fmaxf(fmaf(v.x, qs.x, qb.x), post_lo);
        v.y = fmaxf(fmaf(v.y, qs.y, qb.y), post_lo);
        v.z = fmaxf(fmaf(v.z, qs.z, qb.z), post_lo);
        v.w = fmaxf(fmaf(v.w, qs.w, qb.w), post_lo);
        vout[it] = v;
    }
    // ... and the 16 stores leave back to back.  (The empty asm pins every value and address in a VGPR here: LLVM otherwise sinks
    // the arithmetic into the `if (ok)` blocks of the stores, which puts the load waits right back between them.)
#pragma unroll
    for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(vout[it]), "+v"(yoffs[it]));
    __builtin_amdgcn_sched_barrier(0);
#if HVN_TRACE_FINE
    if (p.dbg) t_f[2] = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if constexpr (ABL != 4) {
#if HVN_NT
            if (oks[it]) __builtin_nontemporal_store(vout[it], (f32x4 *)(p.y + yoffs[it]));
#else
            if (oks[it]) *(f32x4 *)(p.y + yoffs[it]) = vout[it];
#endif
        } else if (vout[it].x == 12345.678f && oks[it]) *(f32x4 *)(p.y + yoffs[it]) = vout[it];   // keeps the math alive, stores nothing
    }
#if HVN_TRACE_FINE
    if (p.dbg) t_f[3] = __builtin_readcyclecounter();        // stores issued (not yet acknowledged)
#endif
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores of this thread have left
        unsigned long long *d = p.dbg + (unsigned long long)HVN_TRACE_WORDS * (blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y);
#if HVN_TRACE_FINE
        d[4] = t_f[0];
        d[5] = t_f[1];
        d[6] = t_f[2];
        d[7] = t_f[3];
#endif
        d[0] = t_start;
        d[1] = t_kend;
        d[2] = __builtin_readcyclecounter();
        d[3] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | ((32 - 1) << 11)) |
               ((unsigned long long)__builtin_amdgcn_s_getreg(6 | (0 << 6) | ((8 - 1) << 11)) << 32);
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PADDED, int ABL = 0, bool GROUPED = false, bool HAS_PRE = true, bool HAS_X2 = false>
static int launch_conv(const ConvArgs &a, hipStream_t stream)
{
    ConvArgs p = a;
    static int stagger = -1;
    if (stagger < 0) {
        const char *e = getenv("HVN_STAGGER");   // experiment switch (profiles/r02_experiments.md section 7): 0 = off (default)
        stagger = e ? atoi(e) : 0;
    }
    p.stagger = stagger;
    static unsigned long long *dbg_buf = nullptr;
    static int dbg_on = -1;
    if (dbg_on < 0) {
        const char *e = getenv("HVN_CONV_TRACE");   // path of a file to dump per-workgroup timestamps of the LAST launch into
        dbg_on = e ? 1 : 0;
        if (dbg_on) hipMalloc(&dbg_buf, HVN_TRACE_WORDS * 8 * (size_t)(1 << 20));
    }
    p.dbg = dbg_on ? dbg_buf : nullptr;
    p.m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = (p.Cout + BN - 1) / BN;
    constexpr size_t stage_fl = (size_t)2 * (BM + BN) * LDS_LD, ep_fl = (size_t)BM * (BN + 4);   // staging buffers / epilogue tile (floats)
    const size_t lds = (stage_fl > ep_fl ? stage_fl : ep_fl) * sizeof(float);
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_conv_igemm_f32<BM, BN, WAVES_M, WAVES_N, PADDED, ABL, GROUPED, HAS_PRE, HAS_X2>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long groups = (p.m_tiles + 7) / 8;
    const long grid = groups * 8 * p.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, p.nbatch > 1 ? p.nbatch : 1), dim3(256), lds, stream, p);
    if (dbg_on && grid * (p.nbatch > 1 ? p.nbatch : 1) <= (1 << 20)) {   // experiment mode only: synchronous dump
        hipStreamSynchronize(stream);
        const size_t n = (size_t)grid * (p.nbatch > 1 ? p.nbatch : 1);
        std::vector<unsigned long long> h(HVN_TRACE_WORDS * n);
        hipMemcpy(h.data(), dbg_buf, 8 * HVN_TRACE_WORDS * n, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(getenv("HVN_CONV_TRACE"), "wb")) {
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hvn_launch_dense_grouped(const ConvArgs &a, hipStream_t stream);

int hvn_launch_conv(const ConvArgs &a, int tile_n, hipStream_t stream)
{
    if (a.Cin % BK != 0 || a.Cin <= 0 || a.Cout % 4 != 0) return -1;
    if (a.M <= 0 || a.M >= (1L << 31) - 256) return -1;  // 32-bit pixel index arithmetic in the kernel
    // 32-bit per-thread byte offsets relative to the sample of the tile's first row; 2^31 and beyond is the "load zeros" range of the
    // buffer descriptor.  A tile of bm rows reaches (HoWo + bm - 2) / HoWo samples ahead -- ONE when a sample holds a tile's worth of
    // pixels, FOUR for a Winograd-domain product with 36 tiles per sample: with the arena's 0.5 GB sample stride that is beyond the
    // reach, and rows would silently read zeros (round 4: the F(6x6,3x3) product of d3 in 'fast' mode).  Refused here instead.
    const int bm = tile_n == 320 ? 256 : 128;
    const long howo = (long)a.Ho * a.Wo;
    if (howo <= 0) return -1;
    const long ahead = (howo + bm - 2) / howo;
    const long span = ahead * a.xsn + (long)(a.H + a.KH) * a.xsy + (long)(a.W + a.KW) * a.xsx;
    if (span < 0 || span * 4 >= (1L << 31)) return -1;
    if (a.x2 && (ahead * a.x2sn + (long)a.H * a.x2sy * a.stride2) * 4 >= (1L << 31)) return -1;
    if ((long)(a.Cout + 128) * a.KH * a.KW * a.Cin * 4 >= (1L << 32)) return -1;
    // "padded" = some tap of some output pixel falls outside the input window
    const bool padded = a.pad_t > 0 || a.pad_l > 0 || (a.Ho - 1) * a.stride - a.pad_t + a.KH > a.H ||
                        (a.Wo - 1) * a.stride - a.pad_l + a.KW > a.W;
    if (padded && a.pre_s) return -1;  // zero padding is produced by the load, before a prologue could run
    static int abl = -1;
    if (abl < 0) {
        const char *e = getenv("HVN_CONV_ABLATE");
        abl = e ? atoi(e) : 0;
    }
    if (a.x2) {  // fused shortcut: 1x1, no prologue, no padding (validated by the caller)
        if (padded || a.Cin2 % BK) return -1;
        if (tile_n == 128) return launch_conv<128, 128, 2, 2, false, 0, false, false, true>(a, stream);
        if (tile_n == 64) return launch_conv<128, 64, 4, 1, false, 0, false, false, true>(a, stream);
        return -1;
    }
    switch (tile_n) {
    case 128:
        if (abl == 1) return launch_conv<128, 128, 2, 2, true, 1>(a, stream);
        if (abl == 2) return launch_conv<128, 128, 2, 2, true, 2>(a, stream);
        if (abl == 3) return launch_conv<128, 128, 2, 2, true, 3>(a, stream);
        if (abl == 4) return launch_conv<128, 128, 2, 2, true, 4>(a, stream);
        if (abl == 6) return launch_conv<128, 128, 2, 2, true, 6>(a, stream);
        if (abl == 5) return launch_conv<128, 128, 2, 2, true, 0>(a, stream);   // the same instantiation without ablation (PADDED + prologue code paths on): the baseline of the ablation series
        if (!a.pre_s && !getenv("HVN_NO_RAWSTORE"))
            return padded ? launch_conv<128, 128, 2, 2, true, 0, false, false>(a, stream) : launch_conv<128, 128, 2, 2, false, 0, false, false>(a, stream);
        return padded ? launch_conv<128, 128, 2, 2, true>(a, stream) : launch_conv<128, 128, 2, 2, false>(a, stream);
    case 64:
        if (!a.pre_s && !getenv("HVN_NO_RAWSTORE"))
            return padded ? launch_conv<128, 64, 4, 1, true, 0, false, false>(a, stream) : launch_conv<128, 64, 4, 1, false, 0, false, false>(a, stream);
        return padded ? launch_conv<128, 64, 4, 1, true>(a, stream) : launch_conv<128, 64, 4, 1, false>(a, stream);
    case 320:
        // 256 x 64 tiles for the 64-channel layers (d0's 3x3 convs, u1.conva's Winograd products): a wave owns 64 x 64 like in the
        // 128 x 128 tile, so a k-step carries 64 MFMAs per wave instead of 32 between two barriers (tile_n = 64 | 0x100; chosen per
        // launch shape by Engine.autotune_tiles; same k order per output element, same bits)
        if (!a.pre_s && !getenv("HVN_NO_RAWSTORE"))
            return padded ? launch_conv<256, 64, 4, 1, true, 0, false, false>(a, stream) : launch_conv<256, 64, 4, 1, false, 0, false, false>(a, stream);
        return padded ? launch_conv<256, 64, 4, 1, true>(a, stream) : launch_conv<256, 64, 4, 1, false>(a, stream);
    case 32:
        // dense-unit conv2: patch-staged kernel (no per-tap restaging); HVN_NO_DENSE_KERNEL=1 falls back to the generic grouped path
        if (a.groups == 4 && a.Cin == 128 && a.Cout == 32 && a.stride == 1 && !padded && !a.pre_s && !a.res && !a.post_s && a.nbatch <= 1 &&
            a.KH == a.KW && (a.KH == 5 || a.KH == 3) && !getenv("HVN_NO_DENSE_KERNEL"))
            return hvn_launch_dense_grouped(a, stream);
        if (a.groups == 4 && a.Cin == 128 && a.Cout == 32 && !getenv("HVN_NO_GROUPED"))
            return padded ? launch_conv<128, 32, 4, 1, true, 0, true>(a, stream) : launch_conv<128, 32, 4, 1, false, 0, true>(a, stream);
        if (a.groups != 1 && !(a.groups == 4 && a.Cin == 128 && a.Cout == 32)) return -1;
        return padded ? launch_conv<128, 32, 4, 1, true>(a, stream) : launch_conv<128, 32, 4, 1, false>(a, stream);
    default: return -1;
    }
}

// =========================================================================================
// Dense-unit conv2 (net_utils.py:114-125): k x k (5 or 3), stride 1, valid, 128 -> 32 channels in 4 groups (32 -> 8).
// The generic kernel re-stages its 128-pixel A tile for each of the 4 x 25 (slab, tap) k-steps and does only 16 small
// MFMAs per step; here a workgroup stages the (8+k-1) x (16+k-1) input patch of TWO groups once (64 channels, 65 KB,
// pixel pitch 68 floats: conflict-free ds_read_b128 for 16 consecutive pixels) and walks all taps out of LDS with no
// further barrier.  Wave = (group, half of the 8 x 16 pixel tile); per tap 4 pixel rows x 8 v_mfma_f32_16x16x4_f32
// (K = 32 channels; the k-labelling is permuted so that a lane reads its 8 channels with two ds_read_b128), the
// 8 x 32 weights of the tap come straight from the packed block-diagonal weights ([32][4][taps][32]) as two 16-byte
// loads per lane, prefetched one tap ahead.  Half of each MFMA's 16 output columns are padding (8 channels per group).
// =========================================================================================
#define DG_TH 8
#define DG_TW 16
#define DG_PP 68   // pixel pitch in floats (64 channels + 4)
template <int KS>
__global__ __launch_bounds__(256, 2) void hvn_dense_grouped_f32(const ConvArgs p, int tiles_x, int tiles_y)
{
    constexpr int PH = DG_TH + KS - 1, PW = DG_TW + KS - 1;
    extern __shared__ __attribute__((aligned(16))) float dg_patch[];   // [PH][PW][DG_PP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    const int gp = bid & 1;            // group pair: groups 2gp, 2gp+1
    bid >>= 1;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int oy0 = ty * DG_TH, ox0 = tx * DG_TW;
    // ---- stage the patch of this group pair (64 channels) -------------------------------------------
    const float *xin = p.x + (long)n * p.xsn + gp * 64;
    for (int i = tid; i < PH * PW * 16; i += 256) {
        const int c4 = i & 15, pix = i >> 4;
        const int py = pix / PW, px = pix - py * PW;
        const int iy = oy0 + py, ix = ox0 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy < p.H && ix < p.W) v = *(const f32x4 *)(xin + (long)iy * p.xsy + (long)ix * p.xsx + c4 * 4);
        *(f32x4 *)(dg_patch + pix * DG_PP + c4 * 4) = v;
    }
    const int gl = wave & 1, half = wave >> 1;   // group inside the pair, tile rows 4*half .. 4*half+3
    const int g = 2 * gp + gl;
    // B: lane (n = l15, q): weights of output channel 8g + n (n < 8), input channels 8q .. 8q+7 of the group's slab
    const bool bvalid = l15 < 8;
    const float *wrow = p.w + ((long)(8 * g + (bvalid ? l15 : 0)) * 4 + g) * (KS * KS) * 32 + 8 * q;
    // weights of a tap = 8 floats per lane; a tap is only ~1000 matrix cycles, less than an L2 round trip, so the loads run
    // three taps ahead through a ring of four register slots (the tap loop is fully unrolled: static slot indices)
    constexpr int TAPS = KS * KS;
    f32x4 wb[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) wb[t][0] = wb[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (bvalid) {
#pragma unroll
        for (int t = 0; t < 3 && t < TAPS; ++t) {
            wb[t][0] = *(const f32x4 *)(wrow + t * 32);
            wb[t][1] = *(const f32x4 *)(wrow + t * 32 + 4);
        }
    }
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // A: lane (pixel column l15 of tile row 4*half + m, q): channels 32*gl + 8q .. +7 at the tap-shifted pixel
    const float *abase = dg_patch + ((4 * half) * PW + l15) * DG_PP + 32 * gl + 8 * q;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        const int r = tap / KS, s2 = tap % KS;
        if (bvalid && tap + 3 < TAPS) {
            wb[(tap + 3) & 3][0] = *(const f32x4 *)(wrow + (tap + 3) * 32);
            wb[(tap + 3) & 3][1] = *(const f32x4 *)(wrow + (tap + 3) * 32 + 4);
        }
        const float *a = abase + (r * PW + s2) * DG_PP;
        f32x4 a0[4], a1[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            a0[m] = *(const f32x4 *)(a + m * PW * DG_PP);
            a1[m] = *(const f32x4 *)(a + m * PW * DG_PP + 4);
        }
        const f32x4 b0 = wb[tap & 3][0], b1 = wb[tap & 3][1];
        // channel step outermost: the four pixel rows' accumulators alternate, no MFMA waits on the previous one
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][e], b0[e], acc[m], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[m][e], b1[e], acc[m], 0, 0, 0);
    }
    // D[m][n]: lane holds n = l15 (output channel 8g + n, n < 8), pixel columns 4q + i (i = 0..3) of tile row 4*half + m
    if (bvalid) {
        const int co = 8 * g + l15;
        const float bias = p.bias ? p.bias[co] : 0.f;
        const float lo = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int oy = oy0 + 4 * half + m;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ox = ox0 + 4 * q + i;
                if (oy < p.Ho && ox < p.Wo) p.y[(long)n * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + co] = fmaxf(acc[m][i] + bias, lo);
            }
        }
    }
}

template <int KS>
static int launch_dense_grouped(const ConvArgs &a, hipStream_t stream)
{
    const int tiles_x = (a.Wo + DG_TW - 1) / DG_TW, tiles_y = (a.Ho + DG_TH - 1) / DG_TH;
    const size_t lds = (size_t)(DG_TH + KS - 1) * (DG_TW + KS - 1) * DG_PP * sizeof(float);
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_dense_grouped_f32<KS>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long grid = 2L * tiles_x * tiles_y * a.N;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a, tiles_x, tiles_y);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}


// Second form of the dense-unit conv2: the 16 MFMA output columns hold 8 channels x 2 horizontally adjacent output
// pixels (dx = 0, 1) instead of 8 channels + 8 columns of padding.  A row of the GEMM is a BASE pixel at an even column;
// the reduction runs over KS x (KS+1) shifted input taps with the weights W'[(ty, tx', ch)][(co, dx)] = w[ty][tx' - dx][ch][co]
// (zero where tx' - dx falls outside the filter): KS(KS+1) tap steps produce two pixels, 5/6 (3/4 for 3x3) of the MFMA
// work is useful instead of 1/2.  Tile = 4 rows x 32 columns; wave = (group, two of the rows); lane l15 = base pixel
// column (stride 2 pixels = 8 LDS banks at 16 bytes per lane: conflict-free).
#define DG2_TH 4
#define DG2_TW 32
template <int KS>
__global__ __launch_bounds__(256, 2) void hvn_dense_grouped2_f32(const ConvArgs p, int tiles_x, int tiles_y)
{
    constexpr int PH = DG2_TH + KS - 1, PW = DG2_TW + KS - 1;
    constexpr int TAPS2 = KS * (KS + 1);
    extern __shared__ __attribute__((aligned(16))) float dg_patch[];   // [PH][PW][DG_PP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    const int gp = bid & 1;
    bid >>= 1;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int oy0 = ty * DG2_TH, ox0 = tx * DG2_TW;
    const float *xin = p.x + (long)n * p.xsn + gp * 64;
    for (int i = tid; i < PH * PW * 16; i += 256) {
        const int c4 = i & 15, pix = i >> 4;
        const int py = pix / PW, px = pix - py * PW;
        const int iy = oy0 + py, ix = ox0 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy < p.H && ix < p.W) v = *(const f32x4 *)(xin + (long)iy * p.xsy + (long)ix * p.xsx + c4 * 4);
        *(f32x4 *)(dg_patch + pix * DG_PP + c4 * 4) = v;
    }
    const int gl = wave & 1, half = wave >> 1;   // group inside the pair; tile rows 2*half, 2*half + 1
    const int g = 2 * gp + gl;
    const int co = l15 & 7, dx = l15 >> 3;       // this lane's output column: channel 8g + co of the pixel at base + dx
    const float *wrow = p.w + ((long)(8 * g + co) * 4 + g) * (KS * KS) * 32 + 8 * q;
    // weights of shifted tap t' = (tr, tc'), tc' in [0, KS]: filter tap (tr, tc' - dx) or zero
    auto load_w = [&](int t2, f32x4 &w0, f32x4 &w1) {
        const int tr = t2 / (KS + 1), tc = t2 - tr * (KS + 1) - dx;
        w0 = w1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tc >= 0 && tc < KS) {
            w0 = *(const f32x4 *)(wrow + (tr * KS + tc) * 32);
            w1 = *(const f32x4 *)(wrow + (tr * KS + tc) * 32 + 4);
        }
    };
    f32x4 wb[4][2];
#pragma unroll
    for (int t = 0; t < 3; ++t) load_w(t, wb[t][0], wb[t][1]);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __syncthreads();
    // A: lane (base pixel column 2*l15 of tile row 2*half + m, q): channels 32*gl + 8q .. +7 at the shifted pixel
    const float *abase = dg_patch + ((2 * half) * PW + 2 * l15) * DG_PP + 32 * gl + 8 * q;
#pragma unroll
    for (int t2 = 0; t2 < TAPS2; ++t2) {
        const int tr = t2 / (KS + 1), tc = t2 % (KS + 1);
        if (t2 + 3 < TAPS2) load_w(t2 + 3, wb[(t2 + 3) & 3][0], wb[(t2 + 3) & 3][1]);
        const float *a = abase + (tr * PW + tc) * DG_PP;
        f32x4 a0[2], a1[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            a0[m] = *(const f32x4 *)(a + m * PW * DG_PP);
            a1[m] = *(const f32x4 *)(a + m * PW * DG_PP + 4);
        }
        const f32x4 b0 = wb[t2 & 3][0], b1 = wb[t2 & 3][1];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][e], b0[e], acc[m], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[m][e], b1[e], acc[m], 0, 0, 0);
    }
    // D[m][n]: lane holds column n = l15 = (co, dx); rows = base pixels 4q + i (i = 0..3) of tile row 2*half + m
    const int ch = 8 * g + co;
    const float bias = p.bias ? p.bias[ch] : 0.f;
    const float lo = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int oy = oy0 + 2 * half + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ox = ox0 + 2 * (4 * q + i) + dx;
            if (oy < p.Ho && ox < p.Wo) p.y[(long)n * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + ch] = fmaxf(acc[m][i] + bias, lo);
        }
    }
}

template <int KS>
static int launch_dense_grouped2(const ConvArgs &a, hipStream_t stream)
{
    const int tiles_x = (a.Wo + DG2_TW - 1) / DG2_TW, tiles_y = (a.Ho + DG2_TH - 1) / DG2_TH;
    const size_t lds = (size_t)(DG2_TH + KS - 1) * (DG2_TW + KS - 1) * DG_PP * sizeof(float);
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_dense_grouped2_f32<KS>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long grid = 2L * tiles_x * tiles_y * a.N;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a, tiles_x, tiles_y);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}


// Third form (round 4): form 2's arithmetic on a ROLLING patch.  Forms 1 / 2 stage a tile's whole input patch, compute, and leave: with two
// workgroups per CU the matrix pipe waits for a staging phase about as long as the tap loop (measured 36 - 50 TFLOP/s, 48 % of the
// form's own MFMA bound).  Here a workgroup walks a RUN of vertically consecutive 4 x 32 tiles: consecutive tiles share all but four
// patch rows, so the patch lives in a ring of 8 rows (slot = input row mod 8), only the four NEW rows are fetched per tile (half the
// bytes), and they are fetched into registers WHILE the current tile's 30 tap steps run; they are parked in LDS between two barriers
// once the tile's reads are done.  Same products in the same order per output as form 2 (same bits).
template <int KS>
__global__ __launch_bounds__(256, 2) void hvn_dense_grouped3_f32(const ConvArgs p, int tiles_x, int tiles_y, int run, int runs)
{
    constexpr int PH = DG2_TH + KS - 1, PW = DG2_TW + KS - 1;
    constexpr int TAPS2 = KS * (KS + 1);
    constexpr int NEW = 4 * PW * 16;                 // float4s of the four new patch rows
    constexpr int NPF = (NEW + 255) / 256;           // ... per thread
    extern __shared__ __attribute__((aligned(16))) float dg_patch[];   // ring [8][PW][DG_PP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    const int gp = bid & 1;
    bid >>= 1;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int rn = bid % runs;
    const int n = bid / runs;
    const int ty0 = rn * run, ty1 = min(ty0 + run, tiles_y);
    const int ox0 = tx * DG2_TW;
    const float *xin = p.x + (long)n * p.xsn + gp * 64;
    // ---- the first tile's whole patch -----------------------------------------------------------------
    {
        const int oy0 = ty0 * DG2_TH;
        for (int i = tid; i < PH * PW * 16; i += 256) {
            const int c4 = i & 15, pix = i >> 4;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = oy0 + py, ix = ox0 + px;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (iy < p.H && ix < p.W) v = *(const f32x4 *)(xin + (long)iy * p.xsy + (long)ix * p.xsx + c4 * 4);
            *(f32x4 *)(dg_patch + ((iy & 7) * PW + px) * DG_PP + c4 * 4) = v;
        }
    }
    const int gl = wave & 1, half = wave >> 1;   // group inside the pair; tile rows 2*half, 2*half + 1
    const int g = 2 * gp + gl;
    const int co = l15 & 7, dx = l15 >> 3;
    const float *wrow = p.w + ((long)(8 * g + co) * 4 + g) * (KS * KS) * 32 + 8 * q;
    auto load_w = [&](int t2, f32x4 &w0, f32x4 &w1) {
        const int tr = t2 / (KS + 1), tc = t2 - tr * (KS + 1) - dx;
        w0 = w1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tc >= 0 && tc < KS) {
            w0 = *(const f32x4 *)(wrow + (tr * KS + tc) * 32);
            w1 = *(const f32x4 *)(wrow + (tr * KS + tc) * 32 + 4);
        }
    };
    const int ch = 8 * g + co;
    const float bias = p.bias ? p.bias[ch] : 0.f;
    const float lo = p.relu ? 0.f : -__builtin_inff();
    const float *acol = dg_patch + (2 * l15) * DG_PP + 32 * gl + 8 * q;     // + slot * PW * DG_PP + tc * DG_PP
    __syncthreads();
    for (int ty = ty0; ty < ty1; ++ty) {
        const int oy0 = ty * DG2_TH;
        // the four new rows of the NEXT tile: issued now, parked after this tile's reads
        f32x4 pf[NPF];
        const bool more = ty + 1 < ty1;
        const int ny0 = oy0 + DG2_TH + PH - 4;      // first new input row
        if (more) {
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int i = tid + 256 * j;
                const int c4 = i & 15, pix = i >> 4;
                const int py = pix / PW, px = pix - py * PW;
                const int iy = ny0 + py, ix = ox0 + px;
                pf[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (i < NEW && iy < p.H && ix < p.W) pf[j] = *(const f32x4 *)(xin + (long)iy * p.xsy + (long)ix * p.xsx + c4 * 4);
            }
        }
        f32x4 wb[4][2];
#pragma unroll
        for (int t = 0; t < 3; ++t) load_w(t, wb[t][0], wb[t][1]);
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const int r0 = oy0 + 2 * half;            // input row of this wave's first output row at tap row 0
#pragma unroll
        for (int t2 = 0; t2 < TAPS2; ++t2) {
            const int tr = t2 / (KS + 1), tc = t2 % (KS + 1);
            if (t2 + 3 < TAPS2) load_w(t2 + 3, wb[(t2 + 3) & 3][0], wb[(t2 + 3) & 3][1]);
            f32x4 a0[2], a1[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float *a = acol + (((r0 + m + tr) & 7) * PW + tc) * DG_PP;
                a0[m] = *(const f32x4 *)(a);
                a1[m] = *(const f32x4 *)(a + 4);
            }
            const f32x4 b0 = wb[t2 & 3][0], b1 = wb[t2 & 3][1];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][e], b0[e], acc[m], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[m][e], b1[e], acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = oy0 + 2 * half + m;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ox = ox0 + 2 * (4 * q + i) + dx;
                if (oy < p.Ho && ox < p.Wo) p.y[(long)n * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + ch] = fmaxf(acc[m][i] + bias, lo);
            }
        }
        if (more) {
            __syncthreads();                       // every wave is done with the four rows that leave the ring
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int i = tid + 256 * j;
                const int c4 = i & 15, pix = i >> 4;
                const int py = pix / PW, px = pix - py * PW;
                if (i < NEW) *(f32x4 *)(dg_patch + (((ny0 + py) & 7) * PW + px) * DG_PP + c4 * 4) = pf[j];
            }
            __syncthreads();
        }
    }
}

template <int KS>
static int launch_dense_grouped3(const ConvArgs &a, int run, hipStream_t stream)
{
    const int tiles_x = (a.Wo + DG2_TW - 1) / DG2_TW, tiles_y = (a.Ho + DG2_TH - 1) / DG2_TH;
    const int runs = (tiles_y + run - 1) / run;
    const size_t lds = (size_t)8 * (DG2_TW + KS - 1) * DG_PP * sizeof(float);
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_dense_grouped3_f32<KS>;
    if (hvn_max_lds_once((const void *)kern, (int)lds, attr_done)) return -2;
    const long grid = 2L * tiles_x * runs * a.N;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a, tiles_x, tiles_y, run, runs);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hvn_launch_dense_grouped(const ConvArgs &a, hipStream_t stream)
{
    static int forced = -1;   // HVN_DENSE_FORM=1 | 2 forces a form (A/B runs); default: by the column-tile waste
    if (forced < 0) {
        const char *e = getenv("HVN_DENSE_FORM");
        forced = e ? atoi(e) : 0;
    }
    // form 2 (two pixels per column block, 4 x 32 tiles) does 40 % fewer MFMAs but pads the width to a multiple of 32:
    // measured faster up to ~1.55x padding (widths 58..42 and 30 of the u3 / u2 dense blocks), slower beyond (38, 34)
    // form 3 (rolling patch, runs of 4 tiles) where it was measured faster than form 2: the tall units (58, 54, 50, 52 rows: 1.12 - 1.22x);
    // below that a run-per-workgroup grid no longer fills the chip (46 .. 34 rows: 0.96 - 1.04x, 30 rows: 0.58x) -- profiles/r04_dense_ab.txt
    const int wide = ((a.Wo + 31) / 32 * 32) * 100 <= a.Wo * 155;
    const int form = forced ? forced : (wide ? (a.Ho >= 50 ? 3 : 2) : 1);
    if (form == 3) {            // form 2's arithmetic on a rolling patch, `run` vertically consecutive tiles per workgroup (HVN_DENSE_RUN)
        static int run = -1;
        if (run < 0) {
            const char *e = getenv("HVN_DENSE_RUN");
            run = e ? atoi(e) : 4;
            if (run < 1) run = 1;
        }
        if (a.KH == 5) return launch_dense_grouped3<5>(a, run, stream);
        if (a.KH == 3) return launch_dense_grouped3<3>(a, run, stream);
        return -1;
    }
    if (form == 2) {
        if (a.KH == 5) return launch_dense_grouped2<5>(a, stream);
        if (a.KH == 3) return launch_dense_grouped2<3>(a, stream);
        return -1;
    }
    if (a.KH == 5) return launch_dense_grouped<5>(a, stream);
    if (a.KH == 3) return launch_dense_grouped<3>(a, stream);
    return -1;
}
