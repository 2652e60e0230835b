// hvn_conv_x3g.hip -- the bf16x3 implicit-GEMM convolution of hvn_conv_x3.hip (fp32 operands, fp32 accumulation, products on the gfx950
// bf16 matrix pipe from exact three-way bf16 splits) with BOTH operands staged by LDS-DMA (`buffer_load_dwordx4 ... lds`) and the fp32 ->
// 3 x bf16 split of the activations moved from the staging pass to the FRAGMENT READ.
//
// Why (round 5; hvn_conv_x3.hip measured at 0.36 .. 0.46 of the bf16 peak): its k-step is two barrier-separated phases -- wait for the
// staged registers, split, ds_write, barrier, issue the next loads, MFMAs -- so a wave's matrix work waits on a global-load round trip that
// was given one k-step to complete, on a 100-instruction VALU pass and on the LDS store path (~50 KB of ds_write per k-step and
// workgroup).  Here a k-step is ONE phase:
//   * A: the raw fp32 activation rows (128 B per pixel and k-step: whole cache lines) go global -> LDS by DMA into a ring of NA stages,
//     issued NA-1 k-steps ahead with a COUNTED s_waitcnt (the ring's youngest stage stays in flight across the barrier); no staging
//     registers, no ds_write.  Rows are XOR-swizzled on the SOURCE side (LDS-DMA writes lane-linear): 16-byte piece p of row r sits at
//     piece p ^ ((r >> 1) & 7), so the 16 lanes of a ds_read_b128 group (distinct r mod 16) cover all 64 banks.
//   * B: the pre-split weight planes ([cout_pad][k-step][3][32] bf16, hvn_conv_x3.hip's packing) by DMA into a ring of two stages laid
//     out [plane][row][64 B] with piece p at p ^ ((r >> 2) & 3).
//   * a wave reads its own A fragment as 8 floats per lane (two ds_read_b128), applies the optional prologue BN-ReLU, splits in registers
//     (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): hvn_conv_x3.hip:split3) and feeds the MFMAs -- 44 VALU per fragment triple against
//     12 MFMAs (a 64 x 64 wave tile), issued under the matrix pipe's shadow; ONE barrier per k-step.
// Every output element sums the same partial products in the same order as hvn_conv_igemm_x3 (same k-slot assignment, same MFMA
// sequence per accumulator): the two kernels are BIT-IDENTICAL (tests/test_gpu_x3.py), so the engine may pick per launch shape by time.
//
// Workgroups: BM = 256 pixels x 128 channels, 512 threads (8 waves as 4 x 2, one per CU: A ring of 3 x 32 KB + B ring of 2 x 24 KB = 144 KB
// of the CU's 160 KB; + the prologue's per-channel vectors for K <= 2048), or BM = 128, 256 threads (4 waves as 2 x 2, two per CU: rings of
// two = exactly 80 KB; the prologue's vectors then come from global memory, a k-step ahead, 32 VGPRs).  Reference geometry as hvn_conv_x3.hip: /root/reference/models/hovernet/net_utils.py:155-266, net_desc.py:76-99.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define GK 32               // reduction elements per k-step
#define GBN 128             // output channels per workgroup
#define G_BSTAGE (3 * GBN * 64)   // bytes of one B stage: [plane 3][row 128][64 B]

typedef __attribute__((address_space(3))) void *lds_ptr_t;

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global offsets (voff + the wave-uniform soff; beyond num_records: zeros)
// to the 1 KiB at the wave-uniform LDS address `dst`, lane-linear.  (The builtin exists in the device pass only; hipcc's host pass
// silently drops a kernel whose body names it, and with it the kernel's launch stub.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_ptr_t dst, unsigned voff, int soff)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
#endif
}

// hvn_conv_x3.hip:split3 for 8 lanes' worth of k: x = h + m + l exactly (RNE conversions; both differences exact in fp32)
__device__ __forceinline__ void split3x8(const f32x4 a, const f32x4 b, bf16x8 &h, bf16x8 &m, bf16x8 &l)
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

template <int BM, bool PADDED, bool HAS_PRE, bool HAS_X2, int NTERMS>
__global__ __launch_bounds__(BM * 2, 2) void hvn_conv_igemm_x3g(ConvArgs p)
{
    constexpr int NTHR = BM * 2;                 // 256 | 512
    constexpr int NW = NTHR / 64;                // waves
    // Wave tile: 64 x 64 (TI = TJ = 2: two waves side by side read and split the same A fragment -- 88 VALU per slice and wave next to 24 MFMAs).
    // Round 6 measured the alternative, 32 pixel rows x all 128 channels per wave (HVN_X3G_WN=1, lib.VARIANTS["wn1"]: the split done once, 44
    // VALU per slice, 14 instead of 10 fragment reads, 210 - 248 VGPRs, same per-accumulator MFMA order = same bits): conv launches of a step 42.03
    // vs 41.71 ms, bench 755.4 vs 754.8 tiles/s -- neutral, i.e. the split's VALU is NOT what holds the matrix pipe at 49 % (profiles/r06_x3g_wave_tile_ab.txt).
#ifndef HVN_X3G_WN
#define HVN_X3G_WN 2
#endif
    constexpr int WAVES_N = HVN_X3G_WN;
    constexpr int TI = BM / (NW / WAVES_N) / 32;      // 32-row tiles per wave: 1 | 2
    constexpr int TJ = GBN / WAVES_N / 32;            // 32-column tiles per wave: 4 | 2
    static_assert(TI * TJ == 4, "four accumulators per wave");
    constexpr int NA = BM == 256 ? 3 : 2;        // A ring depth (stages)
    constexpr int A_STAGE = BM * 128;            // bytes: [BM rows][32 floats]
    constexpr int GA = A_STAGE / 1024 / NW;      // LDS-DMA instructions per wave and A stage (1 KiB each): 4
    constexpr int GB = G_BSTAGE / 1024 / NW;     // per B stage: 6 | 3
    constexpr int B_OFF = NA * A_STAGE;
    constexpr int PRE_OFF = B_OFF + 2 * G_BSTAGE;
    constexpr int EP_LD = GBN + 4;
    // the prologue's per-channel vectors: parked in LDS where the rings leave room (one 256-row workgroup per CU), read from global
    // memory a k-step ahead where they do not (two 128-row workgroups per CU fill the 160 KB exactly)
    constexpr bool PRE_LDS = HAS_PRE && BM == 256, PRE_GLB = HAS_PRE && BM == 128;
    static_assert(GA == 4, "the counted s_waitcnt below leaves exactly one A stage (4 DMA instructions per wave) in flight");
    static_assert(NTERMS == 9 || NTERMS == 6, "nine exact partial products, or the six that carry > 2^-24 of the product");
    extern __shared__ __attribute__((aligned(16))) unsigned char gs[];

    const uint16_t *pw = (const uint16_t *)p.w;
    if (p.nbatch > 1) {  // batched launch: one of nbatch independent problems per blockIdx.y (Winograd transform positions)
        p.x += (long)blockIdx.y * p.xb;
        pw += (long)blockIdx.y * p.wb;
        p.y += (long)blockIdx.y * p.yb;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int n0 = n_tile * GBN;
    const unsigned M = (unsigned)p.M;

    // ---- A staging: DMA instruction jj of this wave fills LDS slots (wave GA + jj) 64 + lane of a stage; slot = row * 8 + physical piece,
    //      and holds the row's logical piece phys ^ ((row >> 1) & 7).  Per-lane byte offset (loop-invariant) + the k-step's offset in an SGPR;
    //      an offset beyond num_records loads zeros (padding taps, rows past the batch) -----------------------------------------------
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const unsigned n_blk = m0 / HoWo;
    const long padoff = (long)p.pad_t * p.xsy + (long)p.pad_l * p.xsx;
    constexpr unsigned OOB = 0x80000000u;
    unsigned a_voff[GA], a2_voff[GA];
    int a_iy[GA], a_ix[GA];
#pragma unroll
    for (int jj = 0; jj < GA; ++jj) {
        const int row = (wave * GA + jj) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        const unsigned m = m0 + row;
        const bool ok = m < M;
        const unsigned mm = ok ? m : m0;
        const unsigned n = mm / HoWo;
        const unsigned rem = mm - n * HoWo;
        const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
        a_iy[jj] = ok ? (int)oy * p.stride - p.pad_t : -(1 << 28);
        a_ix[jj] = ok ? (int)ox * p.stride - p.pad_l : -(1 << 28);
        a_voff[jj] = ok ? (unsigned)(((long)(n - n_blk) * p.xsn + (long)(oy * p.stride) * p.xsy + (long)(ox * p.stride) * p.xsx + piece * 4) * 4) : OOB;
        a2_voff[jj] = OOB;
        if constexpr (HAS_X2)
            a2_voff[jj] = ok ? (unsigned)(((long)(n - n_blk) * p.x2sn + (long)(oy * p.stride2) * p.x2sy + (long)(ox * p.stride2) * p.x2sx + piece * 4) * 4) : OOB;
    }
    const float *xblk = p.x + (long)n_blk * p.xsn - padoff;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)xblk, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)pw, 0, 0x7fffffff, 0x00020000);
    const float *x2blk = HAS_X2 ? p.x2 + (long)n_blk * p.x2sn : p.x;
    const __amdgpu_buffer_rsrc_t rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc((void *)x2blk, 0, 0x7fffffff, 0x00020000);
    const int kchunks = p.Cin / GK;
    const int KT1 = p.KH * p.KW * kchunks;
    const int KT = KT1 + (HAS_X2 ? p.Cin2 / GK : 0);
    // ---- B staging: DMA instruction t = wave GB + jj fills slots t 64 + lane; slot = plane * 512 + row * 4 + physical piece -----------
    unsigned w_voff[GB];
#pragma unroll
    for (int jj = 0; jj < GB; ++jj) {
        const int t = wave * GB + jj;
        const int plane = t >> 3;
        const int row = (t & 7) * 16 + (lane >> 2);
        const int piece = (lane & 3) ^ ((row >> 2) & 3);
        w_voff[jj] = (unsigned)((long)(n0 + row) * KT * 192 + plane * 64 + piece * 16);
    }

    // ---- optional prologue vectors (pre-activation BN: relu(x * s + b) per input channel) parked in LDS once: [2][Cin] floats -------
    if constexpr (PRE_LDS) {
        float *pre = (float *)(gs + PRE_OFF);
        for (int c = tid * 4; c < p.Cin; c += NTHR * 4) {
            *(f32x4 *)(pre + c) = *(const f32x4 *)(p.pre_s + c);
            *(f32x4 *)(pre + p.Cin + c) = *(const f32x4 *)(p.pre_b + c);
        }
    }

    // PRE_GLB: scale / shift of the 8 channels this lane feeds an MFMA with, for both 16-deep slices of ONE k-step (32 VGPRs), requested
    // at the top of the k-step before (ahead of that step's DMAs: vector-memory operations complete in issue order) and touched before
    // its barrier, where everything outstanding is waited for anyway -- the compiler then knows them complete and adds no wait of its own
    // behind the DMAs it cannot count on (they are conditional)
    f32x4 pvs[2][2], pvb[2][2];
    const __amdgpu_buffer_rsrc_t rsrc_ps = __builtin_amdgcn_make_buffer_rsrc((void *)(PRE_GLB ? p.pre_s : p.x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_pb = __builtin_amdgcn_make_buffer_rsrc((void *)(PRE_GLB ? p.pre_b : p.x), 0, 0x7fffffff, 0x00020000);
    auto load_pre = [&](int slab) {
        if constexpr (PRE_GLB) {
            const int so = __builtin_amdgcn_readfirstlane(slab * (GK * 4));
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    pvs[q][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_ps, (unsigned)(q * 64 + lh * 32 + h * 16), so, 0));
                    pvb[q][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_pb, (unsigned)(q * 64 + lh * 32 + h * 16), so, 0));
                }
        }
    };
    auto touch_pre = [&]() {
        if constexpr (PRE_GLB) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(pvs[q][h]), "+v"(pvb[q][h]));
        }
    };
    load_pre(0);

    // ---- epilogue coordinates: thread = one 16-byte column piece of rows erow0 + RPP it -------------------------------------------
    constexpr int CH = GBN / 4;           // float4 chunks per row
    constexpr int RPP = NTHR / CH;        // rows per pass
    constexpr int NIT = BM / RPP;         // 16
    const int ecol = (tid % CH) * 4;
    const int erow0 = tid / CH;
    const int co = n0 + ecol;
    const bool cok = co < p.Cout;         // Cout is a multiple of 4 (validated on the host)
    const bool has_res = p.res != nullptr, has_post = p.post_s != nullptr;

    int ld_r = 0, ld_s = 0, ld_c = 0;  // tap row / col / channel slab of the NEXT A stage to issue
    auto issue_a = [&](int kt) {
        // (readfirstlane: the offset is wave-uniform by construction; without it hipcc keeps the slab counter in a VGPR and wraps every
        //  DMA in a waterfall loop)
        int a_soff = __builtin_amdgcn_readfirstlane((int)(((long)ld_r * p.xsy + (long)ld_s * p.xsx + (long)ld_c * GK) * 4));
        const bool second = HAS_X2 && kt >= KT1;
        if constexpr (HAS_X2) a_soff = second ? (kt - KT1) * (GK * 4) : a_soff;
        const int stage = kt % NA;
#pragma unroll
        for (int jj = 0; jj < GA; ++jj) {
            unsigned vo = a_voff[jj];
            if constexpr (PADDED) {
                const bool ok = (unsigned)(a_iy[jj] + ld_r) < (unsigned)p.H && (unsigned)(a_ix[jj] + ld_s) < (unsigned)p.W;
                vo = ok ? vo : OOB;
            }
            lds_ptr_t dst = (lds_ptr_t)(gs + stage * A_STAGE + (wave * GA + jj) * 1024);
            if constexpr (HAS_X2) {
                if (second)
                    dma16(rsrc_a2, dst, a2_voff[jj], a_soff);
                else
                    dma16(rsrc_a, dst, vo, a_soff);
            } else
                dma16(rsrc_a, dst, vo, a_soff);
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
        const int w_soff = kt * 192;
        const int stage = kt & 1;
#pragma unroll
        for (int jj = 0; jj < GB; ++jj) {
            lds_ptr_t dst = (lds_ptr_t)(gs + B_OFF + stage * G_BSTAGE + (wave * GB + jj) * 1024);
            dma16(rsrc_w, dst, w_voff[jj], w_soff);
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: lane (l31, lh) feeds row l31 of a 32-row tile, k = 16 q + 8 lh .. + 7
    const int akey = (l31 >> 1) & 7, bkey = (l31 >> 2) & 3;
    const unsigned a_row = (unsigned)((wm * (32 * TI) + l31) * 128);
    const unsigned b_row = (unsigned)((wn * (32 * TJ) + l31) * 64);
    const int taps = p.KH * p.KW;
    struct Frag {
        bf16x8 a[TI][3], b[TJ][3];
    };
    // one 16-deep slice (k-step kt, half q) of this wave's operands: A rows read raw, pre-activated and split here; B planes as stored
    auto prep = [&](Frag &f, int kt, int q, int c_slab) {
        const unsigned char *as = gs + (kt % NA) * A_STAGE + a_row;
        const unsigned char *bs = gs + B_OFF + (kt & 1) * G_BSTAGE + b_row;
        f32x4 v0[TI], v1[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            v0[i] = *(const f32x4 *)(as + i * 32 * 128 + (((4 * q + 2 * lh) ^ akey) << 4));
            v1[i] = *(const f32x4 *)(as + i * 32 * 128 + (((4 * q + 2 * lh + 1) ^ akey) << 4));
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                f.b[j][pl] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(bs + pl * (GBN * 64) + j * 32 * 64 + (((2 * q + lh) ^ bkey) << 4)));
        if constexpr (HAS_PRE) {
            f32x4 ps0, ps1, pb0, pb1;
            if constexpr (PRE_LDS) {
                const float *pre = (const float *)(gs + PRE_OFF) + c_slab * GK + q * 16 + lh * 8;
                ps0 = *(const f32x4 *)(pre), ps1 = *(const f32x4 *)(pre + 4);
                pb0 = *(const f32x4 *)(pre + p.Cin), pb1 = *(const f32x4 *)(pre + p.Cin + 4);
            } else {
                ps0 = pvs[q][0], ps1 = pvs[q][1];
                pb0 = pvb[q][0], pb1 = pvb[q][1];
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                v0[i].x = fmaxf(fmaf(v0[i].x, ps0.x, pb0.x), 0.f);
                v0[i].y = fmaxf(fmaf(v0[i].y, ps0.y, pb0.y), 0.f);
                v0[i].z = fmaxf(fmaf(v0[i].z, ps0.z, pb0.z), 0.f);
                v0[i].w = fmaxf(fmaf(v0[i].w, ps0.w, pb0.w), 0.f);
                v1[i].x = fmaxf(fmaf(v1[i].x, ps1.x, pb1.x), 0.f);
                v1[i].y = fmaxf(fmaf(v1[i].y, ps1.y, pb1.y), 0.f);
                v1[i].z = fmaxf(fmaf(v1[i].z, ps1.z, pb1.z), 0.f);
                v1[i].w = fmaxf(fmaf(v1[i].w, ps1.w, pb1.w), 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) split3x8(v0[i], v1[i], f.a[i][0], f.a[i][1], f.a[i][2]);
    };
    // smallest partial products first; (plane of a, plane of b) with 0 = high, 2 = low -- hvn_conv_igemm_x3's order, per accumulator.
    // The 4 NTERMS MFMAs of a slice are numbered in that order; mma(f, lo, hi) issues numbers lo .. hi - 1.
    constexpr int NM = 4 * NTERMS;
    auto mma = [&](const Frag &f, int lo, int hi) {
        int idx = 0;
#pragma unroll
        for (int s = 4; s >= 0; --s) {
            if (NTERMS == 6 && s > 2) continue;
#pragma unroll
            for (int pa = 2; pa >= 0; --pa) {
                const int pb = s - pa;
                if (pb < 0 || pb > 2) continue;
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        if (idx >= lo && idx < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][pa], f.b[j][pb], acc[i][j], 0, 0, 0);
                        ++idx;
                    }
            }
        }
    };
    // issue order the scheduler is asked for inside a phase: the slice's LDS reads first, then one MFMA : five VALU -- the next slice's
    // split runs in the shadow of this slice's matrix work
    auto interleave = [&]() {
        if constexpr (PRE_GLB) __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);      // the next k-step's prologue vectors (first phase only)
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * TI + 3 * TJ + (PRE_LDS ? 4 : 0), 0);
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, TI == 1 ? 3 : 5, 0);
        }
    };

    // ---- pipeline.  Unit of work = a 16-deep slice (k-step kt, half q).  While the MFMAs of slice (kt, 0) run, slice (kt, 1) is read and
    //      split; then ONE barrier per k-step, placed where every read of stage kt is done: behind it stage kt + 1 is visible (each wave waited
    //      for its own DMAs of it), the slots of stage kt are free, B(kt + 2) and A(kt + NA) are issued into them, and slice (kt + 1, 0) is
    //      read and split while the MFMAs of (kt, 1) run.  The counted wait leaves the youngest A stage (issued LAST) in flight.
    //      A phase ENDS with the first four MFMAs of the slice it prepared (one per accumulator, each consuming the low plane of an A
    //      fragment, i.e. the whole split): the per-accumulator order is untouched, and the split cannot be sunk out of the phase that
    //      covers it (hipcc moves a value that is only used behind the loop's back edge to the head of the next iteration).
    issue_b(0);
    issue_a(0);
    touch_pre();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // + the prologue vectors' ds_writes
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (KT > 1) {
        issue_b(1);
        issue_a(1);
    }
    if (NA == 3 && KT > 2) issue_a(2);
    Frag f0, f1;
    int c_slab = 0, c_tap = 0;          // channel slab / tap of the k-step whose slices are being prepared (prologue vectors)
    prep(f0, 0, 0, 0);
    mma(f0, 0, 4);
    auto step = [&](int kt, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
        prep(f1, kt, 1, c_slab);
        if constexpr (!LAST) load_pre(c_tap + 1 == taps ? c_slab + 1 : c_slab);      // the next k-step's vectors (this one's are consumed)
        mma(f0, 4, NM);
        mma(f1, 0, 4);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAST) {
            touch_pre();
            // (lgkmcnt(0): this wave's reads of stage kt have RETURNED before any wave may restage its slots)
            if (NA == 3 && kt + 2 < KT)
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < KT) issue_b(kt + 2);
            if (kt + NA < KT) issue_a(kt + NA);
            if (++c_tap == taps) {
                c_tap = 0;
                ++c_slab;
            }
            __builtin_amdgcn_sched_barrier(0);
            prep(f0, kt + 1, 0, c_slab);
            mma(f1, 4, NM);
            mma(f0, 0, 4);
            interleave();
        } else
            mma(f1, 4, NM);
    };
    for (int kt = 0; kt + 1 < KT; ++kt) step(kt, std::false_type{});
    step(KT - 1, std::true_type{});
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                        // every wave is done with the rings: the epilogue tile takes their place

    // ---- epilogue (fp32, as hvn_conv_igemm_x3): accumulators -> LDS tile -> bias / ReLU / + residual / block BN-ReLU -> 16-byte stores
    float *ep = (float *)gs;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (32 * TI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                ep[row * EP_LD + wn * (32 * TJ) + j * 32 + l31] = acc[i][j][r];
            }
    __syncthreads();
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, qs = {1.f, 1.f, 1.f, 1.f}, qb = bias;
    if (cok) {
        if (p.bias) bias = *(const f32x4 *)(p.bias + co);
        if (has_post) {
            qs = *(const f32x4 *)(p.post_s + co);
            qb = *(const f32x4 *)(p.post_b + co);
        }
    }
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    const float post_lo = has_post ? 0.f : -__builtin_inff();
    // Addresses (round 6): 32-bit byte offsets from the sample of the tile's first row, stepped from row to row with scalar increments, through
    // buffer descriptors -- the epilogue used to form `(long)n * sn + (long)oy * sy + (long)ox * sx` for every residual load and every store:
    // ~140 64-bit multiply-adds and ~600 VALU instructions per thread and tile (hipcc's assembly), four times the arithmetic it serves, issued on
    // the SIMDs the co-resident workgroup's split runs on.  A row past the batch or a column past cout gets the out-of-range offset: the load
    // returns zeros, the store is dropped.
    unsigned e_oy, e_ox, y_off, r_off;
    {
        const unsigned m = m0 + erow0;
        const unsigned e_n = m / HoWo;
        const unsigned rem = m - e_n * HoWo;
        e_oy = rem / (unsigned)p.Wo;
        e_ox = rem - e_oy * (unsigned)p.Wo;
        y_off = (unsigned)(((long)(e_n - n_blk) * p.ysn + (long)e_oy * p.ysy + (long)e_ox * p.ysx + co) * 4);
        r_off = (unsigned)(((long)(e_n - n_blk) * p.rsn + (long)e_oy * p.rsy + (long)e_ox * p.rsx + co) * 4);
    }
    const unsigned y_step = (unsigned)(RPP * p.ysx * 4), y_row = (unsigned)((p.ysy - (long)p.Wo * p.ysx) * 4), y_smp = (unsigned)((p.ysn - (long)p.Ho * p.ysy) * 4);
    const unsigned r_step = (unsigned)(RPP * p.rsx * 4), r_row = (unsigned)((p.rsy - (long)p.Wo * p.rsx) * 4), r_smp = (unsigned)((p.rsn - (long)p.Ho * p.rsy) * 4);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void *)(p.y + (long)n_blk * p.ysn), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r =
        __builtin_amdgcn_make_buffer_rsrc((void *)(has_res ? p.res + (long)n_blk * p.rsn : p.y + (long)n_blk * p.ysn), 0, 0x7fffffff, 0x00020000);
    // two halves of NIT / 2 rows each (register pressure): all residual loads, every value finished, then the stores back to back.
    // Round 6: the epilogue exists in the 8 forms {bias + ReLU | neither} x {residual | none} x {block BN-ReLU | none} and a launch takes the one
    // that holds only ITS operations.  hvn_conv_igemm_x3 always computes max(acc + bias, lo) + res, max(fma(., qs, qb), lo') with bias = res = qb
    // = 0, qs = 1 and lo = lo' = -inf standing in for what is absent: identities for every value that is not -0 (and an accumulator that
    // started at +0 is never -0: x + y is -0 only if both are), so the bits are the same -- but 5 VALU per output element, 320 per thread,
    // issued on SIMDs the co-resident workgroup's split needs (the bf16 chain kernel of this round is where that was measured).
    auto epilogue = [&](auto has_bias_t, auto has_res_t, auto has_post_t) {
        constexpr bool HB = decltype(has_bias_t)::value, HR = decltype(has_res_t)::value, HP = decltype(has_post_t)::value;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            constexpr int HN = NIT / 2;
            f32x4 rall[HN];
            unsigned yoffs[HN];
#pragma unroll
            for (int it = 0; it < HN; ++it) {
                const unsigned m = m0 + erow0 + (half * HN + it) * RPP;
                const bool ok = m < M && cok;
                rall[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (HR && has_res) rall[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, ok ? r_off : OOB, 0, 0));
                yoffs[it] = ok ? y_off : OOB;
                // next row of this thread: RPP pixels on; the wraps add what a row / a sample is longer than its pixels
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
            f32x4 vout[HN];
#pragma unroll
            for (int it = 0; it < HN; ++it) {
                f32x4 v = *(const f32x4 *)(ep + (erow0 + (half * HN + it) * RPP) * EP_LD + ecol);
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
#pragma unroll
            for (int it = 0; it < HN; ++it) asm volatile("" : "+v"(vout[it]), "+v"(yoffs[it]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < HN; ++it) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vout[it]), rsrc_y, yoffs[it], 0, 0);
        }
    };
    const bool hb = p.bias != nullptr || p.relu;          // (a ReLU without a bias keeps the full first stage: max(acc + 0, 0))
    using T = std::true_type;
    using F = std::false_type;
#if defined(HVN_X3G_FULL_EPI) && HVN_X3G_FULL_EPI
    epilogue(T{}, T{}, T{});             // A/B build (lib.VARIANTS["fullepi"]): every operation, absent operands as identities
    return;
#endif
    if (hb) {
        if (has_res) {
            if (has_post) epilogue(T{}, T{}, T{}); else epilogue(T{}, T{}, F{});
        } else {
            if (has_post) epilogue(T{}, F{}, T{}); else epilogue(T{}, F{}, F{});
        }
    } else {
        if (has_res) {
            if (has_post) epilogue(F{}, T{}, T{}); else epilogue(F{}, T{}, F{});
        } else {
            if (has_post) epilogue(F{}, F{}, T{}); else epilogue(F{}, F{}, F{});
        }
    }
}

template <int BM, bool PADDED, bool HAS_PRE, bool HAS_X2, int NTERMS>
static int launch_x3g(const ConvArgs &a, hipStream_t stream)
{
    ConvArgs p = a;
    p.m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = (p.Cout + GBN - 1) / GBN;
    constexpr int NA = BM == 256 ? 3 : 2;
    const size_t stage_b = (size_t)NA * BM * 128 + 2 * G_BSTAGE + (HAS_PRE && BM == 256 ? (size_t)2 * a.Cin * 4 : 0), ep_b = (size_t)BM * (GBN + 4) * 4;
    const size_t lds = stage_b > ep_b ? stage_b : ep_b;
    if (lds > 160 * 1024) return -1;
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = hvn_conv_igemm_x3g<BM, PADDED, HAS_PRE, HAS_X2, NTERMS>;
    if (hvn_max_lds_once((const void *)kern, 160 * 1024, attr_done)) return -2;
    const long groups = (p.m_tiles + 7) / 8;
    const long grid = groups * 8 * p.n_tiles;
    if (grid <= 0 || grid > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, p.nbatch > 1 ? p.nbatch : 1), dim3(BM * 2), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int BM, int NTERMS>
static int dispatch_x3g(const ConvArgs &a, bool padded, hipStream_t stream)
{
    if (a.x2) return launch_x3g<BM, false, false, true, NTERMS>(a, stream);
    if (a.pre_s) return launch_x3g<BM, false, true, false, NTERMS>(a, stream);
    return padded ? launch_x3g<BM, true, false, false, NTERMS>(a, stream) : launch_x3g<BM, false, false, false, NTERMS>(a, stream);
}

// Does this launch have an LDS-DMA form?  128-wide column tiles only (the planes of a narrower layer are padded to 64 rows).
int hvn_conv_x3g_supported(const ConvArgs &a, int bm)
{
    if (bm != 256 && bm != 128) return 0;
    if (a.Cout < 128 || a.groups > 1) return 0;
    if (a.pre_s && bm == 256 && ((size_t)3 * bm * 128 + 2 * G_BSTAGE + (size_t)2 * a.Cin * 4 > 160 * 1024)) return 0;
    return 1;
}

// As hvn_launch_conv_x3 (same operands, same packing, same bits), 256 | 128 pixels x 128 channels per workgroup.
int hvn_launch_conv_x3g(const ConvArgs &a, int bm, int terms, hipStream_t stream)
{
    if (!hvn_conv_x3g_supported(a, bm)) return -1;
    if (a.Cin % GK != 0 || a.Cin <= 0 || a.Cout % 4 != 0) return -1;
    if (a.M <= 0 || a.M >= (1L << 31) - 512) return -1;
    // a bm-row tile reaches (HoWo + bm - 2) / HoWo samples ahead of its first row's sample: 32-bit offsets below 2^31 (hvn_launch_conv)
    const long howo = (long)a.Ho * a.Wo;
    if (howo <= 0) return -1;
    const long ahead = (howo + bm - 2) / howo;
    const long span = ahead * a.xsn + (long)(a.H + a.KH) * a.xsy + (long)(a.W + a.KW) * a.xsx;
    if (span < 0 || span * 4 >= (1L << 31)) return -1;
    if (a.x2 && (ahead * a.x2sn + (long)a.H * a.x2sy * a.stride2) * 4 >= (1L << 31)) return -1;
    // the epilogue's 32-bit offsets into y / res, from the sample of the tile's first row
    if ((ahead * a.ysn + (long)(a.Ho + 1) * a.ysy + (long)a.Wo * a.ysx) * 4 >= (1L << 31)) return -1;
    if (a.res && (ahead * a.rsn + (long)(a.Ho + 1) * a.rsy + (long)a.Wo * a.rsx) * 4 >= (1L << 31)) return -1;
    const long kt = (long)a.KH * a.KW * (a.Cin / GK) + (a.x2 ? a.Cin2 / GK : 0);
    if ((long)(a.Cout + 128) * kt * 192 >= (1L << 31)) return -1;
    const bool padded = a.pad_t > 0 || a.pad_l > 0 || (a.Ho - 1) * a.stride - a.pad_t + a.KH > a.H ||
                        (a.Wo - 1) * a.stride - a.pad_l + a.KW > a.W;
    if (padded && a.pre_s) return -1;
    if (a.x2 && (padded || a.Cin2 % GK || a.pre_s)) return -1;
    if (a.pre_s && (((uintptr_t)a.pre_s | (uintptr_t)a.pre_b) & 15)) return -1;
    if (bm == 256) return terms == 6 ? dispatch_x3g<256, 6>(a, padded, stream) : dispatch_x3g<256, 9>(a, padded, stream);
    return terms == 6 ? dispatch_x3g<128, 6>(a, padded, stream) : dispatch_x3g<128, 9>(a, padded, stream);
}
