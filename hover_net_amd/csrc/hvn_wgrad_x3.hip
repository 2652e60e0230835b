// hvn_wgrad_x3.hip -- the weight gradient of a convolution (hvn_train.hip: hvn_conv_wgrad_f32;
// reference /root/reference/models/hovernet/run_desc.py:84-88 loss.backward()) with its products on the gfx950 bf16 matrix pipe from exact
// three-way bf16 splits of BOTH fp32 operands (hvn_conv_x3.hip: x = h + m + l; six or nine exact partial products per product, fp32
// accumulation):   dW[co][tap][ci] = sum over pixels of dY[pixel][co] * X[pixel shifted by the tap][ci].
//
// The reduction runs over PIXELS, and both operands are channel-contiguous in memory, while an MFMA lane wants 8 consecutive k of one
// row.  The transposition is done in registers, for free: a thread loads the SAME channel quad of 8 consecutive pixels (8 coalesced
// 16-byte loads: a wave reads 512 contiguous bytes per pixel), splits its 32 values, and what it holds per channel and plane is then 8
// bf16 of consecutive pixels -- one 16-byte LDS store that IS a lane's MFMA operand.  LDS per operand: [plane 3][channel row 128][32
// pixels x 2 B], 16-byte piece p of row r at p ^ ((r >> 2) & 3) (hvn_conv_x3g.hip's conflict-free layout); the MFMA loop is
// hvn_conv_igemm_x3's with both fragments read as planes.  Each element is split ONCE per workgroup (no redundancy).
// One tile shape: 128 output x 128 input channels per 256-thread workgroup (the shape of 75 % of the weight-gradient time: round 4's
// rocprofv3 summary); other channel counts stay on hvn_conv_wgrad_f32.  Split-K over pixels + fp32 atomics as there.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "hvn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define WX_PLANE (128 * 64)          // bytes of one plane of one operand: [128 rows][32 pixels x 2 B]
#define WX_OPER (3 * WX_PLANE)

template <int NTERMS>
__global__ __launch_bounds__(256, 2) void hvn_conv_wgrad_x3(const WgradArgs p)
{
    static_assert(NTERMS == 9 || NTERMS == 6, "nine exact partial products, or the six that carry > 2^-24 of the product");
    extern __shared__ __attribute__((aligned(16))) unsigned char ws[];    // [2 operands][3 planes][128][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const float *px = p.x + (long)blockIdx.z * p.xb;      // batched launch: blockIdx.z = one of nbatch independent problems
    const float *pdy = p.dy + (long)blockIdx.z * p.db;
    float *pdw = p.dw + (long)blockIdx.z * p.wb;
    int bid = blockIdx.x;
    const int tm = bid % p.tiles_m;
    bid /= p.tiles_m;
    const int tn = bid % p.tiles_n;
    const int tap = bid / p.tiles_n;
    const int tr = tap / p.KW, tq = tap - tr * p.KW;
    const int m0 = tm * 128, n0 = tn * 128;

    const unsigned R = (unsigned)p.N * p.Ho * p.Wo;
    const unsigned r_begin = blockIdx.y * p.rows_per_split;
    const unsigned r_end = min(R, r_begin + p.rows_per_split);
    if (r_begin >= r_end) return;
    const int steps = (int)((r_end - r_begin + 31) / 32);

    // staging role: waves 0, 1 stage dY (operand 0), waves 2, 3 stage X (operand 1); thread = (channel quad cq, pixel group g of 8)
    const bool is_b = tid >= 128;
    const int cq = tid & 31, g = (tid >> 5) & 3;
    const int col = (is_b ? n0 : m0) + 4 * cq;
    const bool col_ok = is_b || col < p.Cout;            // Cout may end inside the last output tile (Cin is a multiple of 128 here)
    const unsigned HoWo = (unsigned)p.Ho * p.Wo;
    unsigned row0 = r_begin + 8 * g;                      // first of this thread's 8 pixels of the current k-step
    int pn, py, pxx;
    {
        const unsigned rr = row0 < R ? row0 : 0;
        pn = rr / HoWo;
        const unsigned rem = rr - pn * HoWo;
        py = rem / p.Wo;
        pxx = rem - py * p.Wo;
    }
    f32x4 raw[8];
    auto load_global = [&]() {
        int n = pn, y = py, x = pxx;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bool live = row0 + e < r_end && col_ok;
            if (!is_b) {
                if (live) v = *(const f32x4 *)(pdy + (long)n * p.dsn + (long)y * p.dsy + (long)x * p.dsx + col);
            } else {
                const int iy = y * p.stride + tr - p.pad_t, ix = x * p.stride + tq - p.pad_l;
                if (live && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    v = *(const f32x4 *)(px + (long)n * p.xsn + (long)iy * p.xsy + (long)ix * p.xsx + col);
            }
            raw[e] = v;
            if (++x == p.Wo) {
                x = 0;
                if (++y == p.Ho) {
                    y = 0;
                    ++n;
                }
            }
        }
        // advance to the next k-step: 32 pixels further
        row0 += 32;
        pxx += 32;
        while (pxx >= p.Wo) {
            pxx -= p.Wo;
            ++py;
        }
        while (py >= p.Ho) {
            py -= p.Ho;
            ++pn;
        }
    };
    // split the 8 pixels x 4 channels and park them: per channel c and plane, the 8 pixels are one 16-byte piece (piece g of row 4 cq + c)
    unsigned char *my = ws + (is_b ? WX_OPER : 0) + (4 * cq) * 64 + ((g ^ (cq & 3)) << 4);
    auto store_lds = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 h, m, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = raw[e][c];
                const __bf16 hh = (__bf16)x;
                const float r = x - (float)hh;
                const __bf16 mm = (__bf16)r;
                h[e] = hh;
                m[e] = mm;
                l[e] = (__bf16)(r - (float)mm);
            }
            *(u32x4 *)(my + c * 64) = __builtin_bit_cast(u32x4, h);
            *(u32x4 *)(my + c * 64 + WX_PLANE) = __builtin_bit_cast(u32x4, m);
            *(u32x4 *)(my + c * 64 + 2 * WX_PLANE) = __builtin_bit_cast(u32x4, l);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int key = (l31 >> 2) & 3;
    const unsigned char *fa_base = ws + (wm * 64 + l31) * 64;
    const unsigned char *fb_base = ws + WX_OPER + (wn * 64 + l31) * 64;
    auto compute = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 fa[2][3], fb[2][3];
            const int off = ((2 * q + lh) ^ key) << 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fa[i][pl] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(fa_base + pl * WX_PLANE + i * 32 * 64 + off));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fb[j][pl] = __builtin_bit_cast(bf16x8, *(const u32x4 *)(fb_base + pl * WX_PLANE + j * 32 * 64 + off));
            // smallest partial products first; (plane of a, plane of b) with 0 = high, 2 = low
#pragma unroll
            for (int s = 4; s >= 0; --s) {
                if (NTERMS == 6 && s > 2) continue;
#pragma unroll
                for (int pa = 2; pa >= 0; --pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb > 2) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa], fb[j][pb], acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    load_global();
    for (int s = 0; s < steps; ++s) {
        __syncthreads();             // every wave is done reading the previous step out of LDS
        store_lds();
        __syncthreads();
        if (s + 1 < steps) load_global();   // lands under this step's MFMAs
        compute();
    }

    // epilogue: D[m][n] of block (i, j): m = 8*(r/4) + 4*lh + (r%4) -> output channel, n = l31 -> input channel; fp32 atomics (split K)
    // deterministic form (p.part): this split's tile is stored into copy blockIdx.y and hvn_reduce_parts adds the copies in split order.
    // Addresses (round 6, as the conv epilogues): ONE 64-bit element offset per thread -- (first output channel, tap, first input channel) --
    // and 32-bit offsets from it for the 64 elements a lane holds (the row of an output channel is taps x Cin_g floats long); the
    // per-element `((long)co * taps + tap) * Cin_g + ci` was ~800 VALU instructions per thread for 64 stores.
    const int taps = p.KH * p.KW;
    float *pout = p.part ? p.part + (long)blockIdx.y * p.part_stride + (long)blockIdx.z * p.wb : pdw;
    const bool det = p.part != nullptr;
    const int co0 = m0 + wm * 64 + 4 * lh, ci0 = n0 + wn * 64 + l31;
    float *pbase = pout + ((long)co0 * taps + tap) * p.Cin_g + ci0;
    const unsigned row = (unsigned)(taps * p.Cin_g);          // floats between consecutive output channels
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dco = i * 32 + 8 * (r >> 2) + (r & 3);
                if (co0 + dco >= p.Cout) continue;
                float *d = pbase + (unsigned)dco * row + (unsigned)(j * 32);
                if (det)
                    *d = acc[i][j][r];
                else
                    unsafeAtomicAdd(d, acc[i][j][r]);
            }
        }
}

// Which launches have this form: ungrouped, cout >= 128 (a multiple of 32, like the fp32 form it stands in for), cin a multiple of 128,
// and -- the kernel issues 16-byte loads at base + n*sn + y*sy + x*sx + 4*cq -- both views 16-byte aligned with strides that are
// multiples of 4 floats (round-5 advisor: a channel-offset view that is not was accepted here and rejected by every other conv path).
int hvn_wgrad_x3_supported(const WgradArgs &a)
{
    if (!(a.groups <= 1 && a.Cout >= 128 && a.Cout % 32 == 0 && a.Cin % 128 == 0)) return 0;
    if ((((uintptr_t)a.x | (uintptr_t)a.dy) & 15) != 0) return 0;
    if (((a.xsn | a.xsy | a.xsx | a.dsn | a.dsy | a.dsx) & 3) != 0) return 0;
    return 1;
}

int hvn_launch_wgrad_x3(WgradArgs a, int terms, hipStream_t stream)
{
    if (!hvn_wgrad_x3_supported(a) || (terms != 6 && terms != 9)) return -1;
    const size_t lds = 2 * WX_OPER;
    a.tiles_m = (a.Cout + 127) / 128;
    a.tiles_n = a.Cin / 128;
    const long tiles = (long)a.tiles_m * a.tiles_n * a.KH * a.KW;
    const int nb = a.nbatch > 1 ? a.nbatch : 1;
    // the K split of hvn_train.hip:launch_wgrad (the same knobs: the engine's target in want_wgs, HVN_WGRAD_WGS / _MIN_ROWS)
    const long ksplit = hvn_wgrad_split(a, tiles, &a.rows_per_split);
    const long elems = a.nbatch > 1 ? (long)a.nbatch * a.wb : (long)a.Cout * a.KH * a.KW * a.Cin_g;
    if (a.part && ksplit > 1) {
        if (ksplit * elems > a.part_cap) return -4;
        a.part_stride = elems;
    } else
        a.part = nullptr;       // a single split: one writer per element
    static std::atomic<unsigned long long> attr6{0}, attr9{0};
    if (terms == 6) {
        if (hvn_max_lds_once(reinterpret_cast<const void *>(hvn_conv_wgrad_x3<6>), (int)lds, attr6)) return -2;
        hipLaunchKernelGGL(hvn_conv_wgrad_x3<6>, dim3((unsigned)tiles, (unsigned)ksplit, (unsigned)nb), dim3(256), lds, stream, a);
    } else {
        if (hvn_max_lds_once(reinterpret_cast<const void *>(hvn_conv_wgrad_x3<9>), (int)lds, attr9)) return -2;
        hipLaunchKernelGGL(hvn_conv_wgrad_x3<9>, dim3((unsigned)tiles, (unsigned)ksplit, (unsigned)nb), dim3(256), lds, stream, a);
    }
    if (hipGetLastError() != hipSuccess) return -2;
    return a.part ? hvn_launch_reduce_parts(a.dw, a.part, elems, ksplit, elems, stream) : 0;
}
