// hvn_net_ops.hip -- the non-GEMM launches of the network plan (HBM-bound byte movers).
//
//   conv0    /root/reference/models/hovernet/net_desc.py:27-35,103 : uint8 image -> 7x7x3
//            conv (1/255 and BN folded) + ReLU.  K = 147 is too thin for the matrix cores;
//            done on the VALU with the taps streamed through the scalar cache (they are
//            wave-uniform), the uint8 patch staged once in LDS.
//   upadd    net_utils.py:284-294 + net_desc.py:133,136,139 : nearest 2x upsample + skip add
//   head     net_desc.py:62-68 u0.conv : 64 -> {2..} logits + bias, written NCHW
//   predmap  run_desc.py:185-194 : softmax(np)[1], argmax(softmax(tp)), concat
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "hvn_kernels.h"

// ---------------------------------------------------------------------------------------
typedef __bf16 hvn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline float hvn_bf_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ inline float hvn_bf_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ inline uint32_t hvn_pack_bf(float a, float b)
{
    hvn_bf16x2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, h);
}

#define C0_T 16              // output tile edge
#define C0_P (C0_T + 6)      // patch edge
template <typename T>
__global__ __launch_bounds__(256) void hvn_conv0(const Conv0Args p)
{
    __shared__ float patch[C0_P][C0_P * 3 + 2];
    const int tid = threadIdx.x;
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * C0_T, ox0 = blockIdx.x * C0_T;
    const T *img = (const T *)p.img + (long)n * p.isn;
    for (int i = tid; i < C0_P * C0_P * 3; i += 256) {
        const int py = i / (C0_P * 3), pr = i - py * (C0_P * 3);
        const int px = pr / 3, ch = pr - px * 3;
        const int iy = oy0 + py - p.pad, ix = ox0 + px - p.pad;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = (float)img[(long)iy * p.isy + (long)ix * p.isx + (long)ch * p.isc];
        patch[py][pr] = v;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    float acc[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) acc[c] = p.bias[c];
    for (int r = 0; r < 7; ++r) {
        const float *prow = &patch[ty + r][tx * 3];
        for (int s3 = 0; s3 < 21; ++s3) {
            const float v = prow[s3];
            const float *__restrict__ wp = p.w + (r * 21 + s3) * 64;  // wave-uniform -> s_load
#pragma unroll
            for (int c = 0; c < 64; ++c) acc[c] = fmaf(v, wp[c], acc[c]);
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    const float lo = p.relu ? 0.f : -__builtin_inff();
    if (oy < p.Ho && ox < p.Wo) {
        const long yoff = (long)n * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx;
        if (p.out_bf16) {   // bf16 activations (RNE)
            uint16_t *y = (uint16_t *)p.y + yoff;
#pragma unroll
            for (int c = 0; c < 64; c += 4) {
                hvn_bf16x2 h0 = {(__bf16)fmaxf(acc[c], lo), (__bf16)fmaxf(acc[c + 1], lo)};
                hvn_bf16x2 h1 = {(__bf16)fmaxf(acc[c + 2], lo), (__bf16)fmaxf(acc[c + 3], lo)};
                *(uint2 *)(y + c) = make_uint2(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1));
            }
        } else {
            float *y = p.y + yoff;
#pragma unroll
            for (int c = 0; c < 64; c += 4)
                *(float4 *)(y + c) = make_float4(fmaxf(acc[c], lo), fmaxf(acc[c + 1], lo), fmaxf(acc[c + 2], lo), fmaxf(acc[c + 3], lo));
        }
    }
}

// conv0 on the matrix cores: 16x16 output pixels x 64 channels per workgroup, K = 7*7*3 = 147 (+1 zero) walked as 74
// v_mfma_f32_32x32x2_f32 steps per block.  The A operand is gathered straight out of the staged image patch (lane = pixel,
// k -> (tap row, tap col * 3 + ch) is a compile-time offset), B out of the [148][64] taps in LDS; 1.5 LDS reads per MFMA.
// Replaces the VALU kernel above (1.46 ms -> see profiles; HVN_CONV0_VALU=1 keeps the old one for A/B runs).
#define C0_PITCH (C0_P * 3 + 2)
template <typename T>
__global__ __launch_bounds__(256) void hvn_conv0_mfma(const Conv0Args p)
{
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ float patch[C0_P + 1][C0_PITCH];   // + one zero row: k = 147 (the padding of the odd K) reads row py + 7
    __shared__ float wl[148][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * C0_T, ox0 = blockIdx.x * C0_T;
    const T *img = (const T *)p.img + (long)n * p.isn;
    for (int i = tid; i < (C0_P + 1) * C0_P * 3; i += 256) {
        const int py = i / (C0_P * 3), pr = i - py * (C0_P * 3);
        const int px = pr / 3, ch = pr - px * 3;
        const int iy = oy0 + py - p.pad, ix = ox0 + px - p.pad;
        float v = 0.f;
        if (py < C0_P && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = (float)img[(long)iy * p.isy + (long)ix * p.isx + (long)ch * p.isc];
        patch[py][pr] = v;
    }
    for (int i = tid; i < 148 * 16; i += 256) {
        const int k = i >> 4, c4 = i & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < 147) v = *(const float4 *)(p.w + k * 64 + c4 * 4);
        *(float4 *)&wl[k][c4 * 4] = v;
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // wave w: tile rows 4w .. 4w+3; M-block i: rows 4w + 2i + {0, 1}; lane pixel p = l31 -> (row p >> 4, col p & 15)
    const float *abase = &patch[4 * wave + (l31 >> 4)][(l31 & 15) * 3];
    const float *bbase = &wl[lh][l31];
#pragma unroll
    for (int t = 0; t < 74; ++t) {
        const int k0 = 2 * t, k1 = 2 * t + 1;
        const int o0 = (k0 / 21) * C0_PITCH + (k0 % 21), o1 = (k1 / 21) * C0_PITCH + (k1 % 21);   // compile-time
        const int off = lh ? o1 : o0;
        const float a0 = abase[off], a1 = abase[off + 2 * C0_PITCH];
        const float b0 = bbase[k0 * 64], b1 = bbase[k0 * 64 + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    const float lo = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = 32 * j + l31;
        const float bias = p.bias[ch];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int oy = oy0 + 4 * wave + 2 * i + (m >> 4), ox = ox0 + (m & 15);
                if (oy < p.Ho && ox < p.Wo) {
                    const float v = fmaxf(acc[i][j][r] + bias, lo);
                    const long yoff = (long)n * p.ysn + (long)oy * p.ysy + (long)ox * p.ysx + ch;
                    if (p.out_bf16) ((__bf16 *)p.y)[yoff] = (__bf16)v;
                    else p.y[yoff] = v;
                }
            }
    }
}

int hvn_launch_conv0(const Conv0Args &a, hipStream_t stream)
{
    dim3 grid((a.Wo + C0_T - 1) / C0_T, (a.Ho + C0_T - 1) / C0_T, a.N);
    static int valu = -1;
    if (valu < 0) valu = getenv("HVN_CONV0_VALU") ? 1 : 0;
    if (!valu) {
        if (a.is_f32)
            hipLaunchKernelGGL(hvn_conv0_mfma<float>, grid, dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL(hvn_conv0_mfma<uint8_t>, grid, dim3(256), 0, stream, a);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (a.is_f32)
        hipLaunchKernelGGL(hvn_conv0<float>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(hvn_conv0<uint8_t>, grid, dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hvn_upadd(const UpAddArgs p, long total4)
{
    const int c4n = p.C >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        long t = i / c4n;
        const int x = (int)(t % p.W);
        t /= p.W;
        const int y = (int)(t % p.H);
        const int n = (int)(t / p.H);
        const float4 a = *(const float4 *)(p.lo + (long)n * p.lsn + (long)(y >> 1) * p.lsy + (long)(x >> 1) * p.lsx + c4 * 4);
        const float4 b = *(const float4 *)(p.skip + (long)n * p.ssn + (long)y * p.ssy + (long)x * p.ssx + c4 * 4);
        *(float4 *)(p.y + (long)n * p.ysn + (long)y * p.ysy + (long)x * p.ysx + c4 * 4) =
            make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// bf16 activations: 8 channels (16 B) per thread, the sum in fp32
__global__ __launch_bounds__(256) void hvn_upadd_bf16(const UpAddArgs p, long total8)
{
    const int c8n = p.C >> 3;
    const uint16_t *lo = (const uint16_t *)p.lo, *skip = (const uint16_t *)p.skip;
    uint16_t *yo = (uint16_t *)p.y;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        long t = i / c8n;
        const int x = (int)(t % p.W);
        t /= p.W;
        const int y = (int)(t % p.H);
        const int n = (int)(t / p.H);
        const uint4 a = *(const uint4 *)(lo + (long)n * p.lsn + (long)(y >> 1) * p.lsy + (long)(x >> 1) * p.lsx + c8 * 8);
        const uint4 b = *(const uint4 *)(skip + (long)n * p.ssn + (long)y * p.ssy + (long)x * p.ssx + c8 * 8);
        uint4 o;
        o.x = hvn_pack_bf(hvn_bf_lo(a.x) + hvn_bf_lo(b.x), hvn_bf_hi(a.x) + hvn_bf_hi(b.x));
        o.y = hvn_pack_bf(hvn_bf_lo(a.y) + hvn_bf_lo(b.y), hvn_bf_hi(a.y) + hvn_bf_hi(b.y));
        o.z = hvn_pack_bf(hvn_bf_lo(a.z) + hvn_bf_lo(b.z), hvn_bf_hi(a.z) + hvn_bf_hi(b.z));
        o.w = hvn_pack_bf(hvn_bf_lo(a.w) + hvn_bf_lo(b.w), hvn_bf_hi(a.w) + hvn_bf_hi(b.w));
        *(uint4 *)(yo + (long)n * p.ysn + (long)y * p.ysy + (long)x * p.ysx + c8 * 8) = o;
    }
}

int hvn_launch_upadd(const UpAddArgs &a, hipStream_t stream)
{
    if (a.bf16) {
        if (a.C % 8) return -1;
        const long total8 = (long)a.N * a.H * a.W * (a.C / 8);
        long blocks = (total8 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(hvn_upadd_bf16, dim3((unsigned)blocks), dim3(256), 0, stream, a, total8);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (a.C % 4) return -1;
    const long total4 = (long)a.N * a.H * a.W * (a.C / 4);
    long blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(hvn_upadd, dim3((unsigned)blocks), dim3(256), 0, stream, a, total4);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hvn_head(const HeadArgs p, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % p.W);
    long t = i / p.W;
    const int y = (int)(t % p.H);
    const int n = (int)(t / p.H);
    const long xoff = (long)n * p.xsn + (long)y * p.xsy + (long)x * p.xsx;
    float v[64];
    if (p.in_bf16) {
        const uint16_t *src = (const uint16_t *)p.x + xoff;
#pragma unroll
        for (int c = 0; c < 64; c += 8) {
            const uint4 q = *(const uint4 *)(src + c);
            v[c] = hvn_bf_lo(q.x); v[c + 1] = hvn_bf_hi(q.x); v[c + 2] = hvn_bf_lo(q.y); v[c + 3] = hvn_bf_hi(q.y);
            v[c + 4] = hvn_bf_lo(q.z); v[c + 5] = hvn_bf_hi(q.z); v[c + 6] = hvn_bf_lo(q.w); v[c + 7] = hvn_bf_hi(q.w);
        }
    } else {
        const float *src = p.x + xoff;
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
            const float4 q = *(const float4 *)(src + c);
            v[c] = q.x;
            v[c + 1] = q.y;
            v[c + 2] = q.z;
            v[c + 3] = q.w;
        }
    }
    const long plane = (long)p.H * p.W;
    float *dst = p.y + (long)n * p.Cout * plane + (long)y * p.W + x;
    for (int co = 0; co < p.Cout; ++co) {
        const float *__restrict__ w = p.w + co * 64;  // uniform
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 64; ++c) acc = fmaf(v[c], w[c], acc);
        dst[co * plane] = acc + p.bias[co];
    }
}

int hvn_launch_head(const HeadArgs &a, hipStream_t stream)
{
    const long total = (long)a.N * a.H * a.W;
    hipLaunchKernelGGL(hvn_head, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hvn_predmap(const PredMapArgs p, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long plane = (long)p.H * p.W;
    const long n = i / plane, pix = i - n * plane;
    const float l0 = p.np[(n * 2) * plane + pix], l1 = p.np[(n * 2 + 1) * plane + pix];
    // torch softmax: exp(x - max) / sum
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float prob = e1 / (e0 + e1);
    const float h = p.hv[(n * 2) * plane + pix], v = p.hv[(n * 2 + 1) * plane + pix];
    if (p.nr_types > 0) {
        // argmax(softmax(tp)) == argmax(tp) (softmax is monotone); first maximum wins like torch.argmax
        int best = 0;
        float bv = p.tp[(n * p.nr_types) * plane + pix];
        for (int t = 1; t < p.nr_types; ++t) {
            const float q = p.tp[(n * p.nr_types + t) * plane + pix];
            if (q > bv) {
                bv = q;
                best = t;
            }
        }
        *(float4 *)(p.y + i * 4) = make_float4((float)best, prob, h, v);
    } else {
        float *y = p.y + i * 3;
        y[0] = prob;
        y[1] = h;
        y[2] = v;
    }
}

int hvn_launch_predmap(const PredMapArgs &a, hipStream_t stream)
{
    const long total = (long)a.N * a.H * a.W;
    hipLaunchKernelGGL(hvn_predmap, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------
// Winograd F(m x m, r x r) transforms (n = m + r - 1: F(2,5), F(4,5) for the 5x5 decoder convs, F(4,3) for the
// encoder's stride-1 3x3 convs with K >= 256) around the batched GEMM
// (net_desc.py:45,52,59 conva; 51 % of the network's FLOPs).  y = A^T [ (G g G^T) .* (B^T d B) ] A:
// n^2 multiplications per m^2 outputs instead of 25 m^2 (hover_net_amd/winograd.py derives the matrices).
// Both kernels are HBM-bound byte movers: one thread = one tile x VW channels, consecutive threads on
// consecutive channels.  VW = 4 (16-byte accesses) where the n^2 x VW register tile fits, 2 for the 8x8 input tile.
#ifndef HVN_WINO_XCD
#define HVN_WINO_XCD 1             // (lib.py VARIANTS "noxcd": 0, the A/B build)
#endif
template <int VW> struct WVec;
template <> struct WVec<4> { typedef float T __attribute__((ext_vector_type(4))); };
template <> struct WVec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct WVec<1> { typedef float T __attribute__((ext_vector_type(1))); };

template <int MO, int R, int VW>
__global__ __launch_bounds__(256) void hvn_wino_in(const WinoArgs p, long total)
{
    constexpr int NW = MO + R - 1;
    typedef typename WVec<VW>::T VT;
    // Neighbouring tiles share (NW - MO) / NW of their input rows and columns, and workgroups are dealt to the 8 XCDs round robin
    // (blockIdx % 8), each with an L2 of its own: in launch order every XCD would fetch every halo from HBM for itself.  XCD k takes
    // the k-th CONTIGUOUS eighth of the (tile, channel slice) list instead, so a halo is re-read from the L2 that already holds it.
    unsigned bid = blockIdx.x;
#if HVN_WINO_XCD
    {
        const unsigned per = gridDim.x >> 3;
        if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
    }
#endif
    const long i = (long)bid * 256 + threadIdx.x;
    if (i >= total) return;
    const int cvn = p.C / VW;
    const int cv = (int)(i % cvn);
    long t = i / cvn;
    const int T1 = p.ty * p.tx;
    const int tile = (int)(t % T1);
    const int n = (int)(t / T1);
    const int tyi = tile / p.tx, txi = tile - tyi * p.tx;
    const int y0 = MO * tyi - p.pad, x0 = MO * txi - p.pad;
    float bt[NW * NW];
#pragma unroll
    for (int k = 0; k < NW * NW; ++k) bt[k] = p.mat[k];  // wave-uniform -> scalar loads
    const float *src = p.x + (long)n * p.xsn + cv * VW;
    const float *lo = p.lo ? p.lo + (long)n * p.lsn + cv * VW : nullptr;      // fused UpSample2x + skip add: input = lo[y/2][x/2] + x[y][x]
    // tmp[a][j] = sum_i BT[a][i] d[i][j], column by column
    VT tmp[NW][NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        VT d[NW];
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const int yy = y0 + r, xx = x0 + j;
            const bool in = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            VT v = in ? *(const VT *)(src + (long)yy * p.xsy + (long)xx * p.xsx) : (VT)(0.f);
            if (lo && in) v = *(const VT *)(lo + (long)(yy >> 1) * p.lsy + (long)(xx >> 1) * p.lsx) + v;      // the sum hvn_upadd forms: same bits
            d[r] = v;
        }
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            VT s = (VT)(0.f);
#pragma unroll
            for (int r = 0; r < NW; ++r) s = __builtin_elementwise_fma((VT)(bt[a * NW + r]), d[r], s);
            tmp[a][j] = s;
        }
    }
    // V[a][b] = sum_j tmp[a][j] BT[b][j]
    float *dst = p.y + (long)n * p.ysn + (long)tile * p.ysx + cv * VW;
#pragma unroll
    for (int a = 0; a < NW; ++a)
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            VT s = (VT)(0.f);
#pragma unroll
            for (int j = 0; j < NW; ++j) s = __builtin_elementwise_fma((VT)(bt[b * NW + j]), tmp[a][j], s);
            *(VT *)(dst + (long)(a * NW + b) * p.ysy) = s;
        }
}

int hvn_launch_wino_in(const WinoArgs &a, hipStream_t stream)
{
    if (a.m == 2 && a.r == 5) {
        const long total = (long)a.N * a.ty * a.tx * (a.C / 4);
        hipLaunchKernelGGL((hvn_wino_in<2, 5, 4>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else if (a.m == 4 && a.r == 3) {
        const long total = (long)a.N * a.ty * a.tx * (a.C / 4);
        hipLaunchKernelGGL((hvn_wino_in<4, 3, 4>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else if (a.m == 4 && a.r == 5) {
        const long total = (long)a.N * a.ty * a.tx * (a.C / 2);
        hipLaunchKernelGGL((hvn_wino_in<4, 5, 2>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else if (a.m == 6 && a.r == 3) {      // F(6x6, 3x3): the same 8 x 8 transform tile as F(4x4, 5x5), 64 products per 36 outputs
        const long total = (long)a.N * a.ty * a.tx * (a.C / 2);
        hipLaunchKernelGGL((hvn_wino_in<6, 3, 2>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else if (a.m == 6 && a.r == 5) {      // F(6x6, 5x5): 10 x 10 tile, one channel per thread (100 products per 36 outputs)
        const long total = (long)a.N * a.ty * a.tx * a.C;
        hipLaunchKernelGGL((hvn_wino_in<6, 5, 1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else
        return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int MO, int R, int VW>
__global__ __launch_bounds__(256) void hvn_wino_out(const WinoArgs p, long total)
{
    constexpr int NW = MO + R - 1;
    typedef typename WVec<VW>::T VT;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4n = p.C / VW;
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int T1 = p.ty * p.tx;
    const int tile = (int)(t % T1);
    const int n = (int)(t / T1);
    const int tyi = tile / p.tx, txi = tile - tyi * p.tx;
    float at[MO * NW];
#pragma unroll
    for (int k = 0; k < MO * NW; ++k) at[k] = p.mat[k];
    const float *src = p.x + (long)n * p.xsn + (long)tile * p.xsx + c4 * VW;
    // tmp[q][b] = sum_a AT[q][a] M[a][b]
    VT tmp[MO][NW];
#pragma unroll
    for (int b = 0; b < NW; ++b) {
        VT m[NW];
#pragma unroll
        for (int a = 0; a < NW; ++a) m[a] = *(const VT *)(src + (long)(a * NW + b) * p.xsy);
#pragma unroll
        for (int q = 0; q < MO; ++q) {
            VT s = (VT)(0.f);
#pragma unroll
            for (int a = 0; a < NW; ++a) s = __builtin_elementwise_fma((VT)(at[q * NW + a]), m[a], s);
            tmp[q][b] = s;
        }
    }
    VT bias = (VT)(0.f);
    if (p.bias) bias = *(const VT *)(p.bias + c4 * VW);
    const float lo = p.relu ? 0.f : -__builtin_inff();
    float *dst = p.y + (long)n * p.ysn + c4 * VW;
#pragma unroll
    for (int q = 0; q < MO; ++q)        // output row
#pragma unroll
        for (int r = 0; r < MO; ++r) {  // output column
            VT s = bias;
#pragma unroll
            for (int b = 0; b < NW; ++b) s = __builtin_elementwise_fma((VT)(at[r * NW + b]), tmp[q][b], s);
            s = __builtin_elementwise_max(s, (VT)(lo));
            const int oy = MO * tyi + q, ox = MO * txi + r;
            if (oy < p.H && ox < p.W) {  // partial last tile when the output extent is not a multiple of m
                VT *d = (VT *)(dst + (long)oy * p.ysy + (long)ox * p.ysx);
                *d = p.accum ? *d + s : s;
            }
        }
}

int hvn_launch_wino_out(const WinoArgs &a, hipStream_t stream)
{
    if (a.m == 2 && a.r == 5) {
        const long total = (long)a.N * a.ty * a.tx * (a.C / 4);
        hipLaunchKernelGGL((hvn_wino_out<2, 5, 4>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else if (a.m == 4 && a.r == 3) {
        const long total = (long)a.N * a.ty * a.tx * (a.C / 4);
        hipLaunchKernelGGL((hvn_wino_out<4, 3, 4>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else if (a.m == 4 && a.r == 5) {
        const long total = (long)a.N * a.ty * a.tx * (a.C / 2);
        hipLaunchKernelGGL((hvn_wino_out<4, 5, 2>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else if (a.m == 6 && a.r == 3) {      // F(6x6, 3x3): the same 8 x 8 transform tile as F(4x4, 5x5), 64 products per 36 outputs
        const long total = (long)a.N * a.ty * a.tx * (a.C / 2);
        hipLaunchKernelGGL((hvn_wino_out<6, 3, 2>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else if (a.m == 6 && a.r == 5) {      // F(6x6, 5x5): 10 x 10 tile, one channel per thread (100 products per 36 outputs)
        const long total = (long)a.N * a.ty * a.tx * a.C;
        hipLaunchKernelGGL((hvn_wino_out<6, 5, 1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    } else
        return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------
// Patch extraction with numpy "reflect" padding folded in (infer/tile.py:46-94 _prepare_patching + the loader's
// crop, dataloader/infer_loader.py:59-72): out[p][y][x][c] = img[refl(cy[p] + y - pad_t)][refl(cx[p] + x - pad_l)][c]
// with refl the mirror-without-edge-repeat index map (period 2(n-1)).  The source image is uploaded once; the
// (win/step)^2-fold overlap of the patches is produced on the GPU instead of crossing PCIe.
__device__ inline int hvn_reflect(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    int j = i % p;
    if (j < 0) j += p;
    return j < n ? j : p - j;
}

__global__ __launch_bounds__(256) void hvn_extract_patches_k(const uint8_t *img, int H, int W, const int32_t *coords, int win, int pad_t,
                                                              int pad_l, uint8_t *out, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % win);
    long t = i / win;
    const int y = (int)(t % win);
    const int p = (int)(t / win);
    const int sy = hvn_reflect(coords[2 * p] + y - pad_t, H), sx = hvn_reflect(coords[2 * p + 1] + x - pad_l, W);
    const uint8_t *s = img + ((long)sy * W + sx) * 3;
    uint8_t *d = out + i * 3;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
}

int hvn_launch_extract_patches(const uint8_t *img, int H, int W, const int32_t *coords, int P, int win, int pad_t, int pad_l, uint8_t *out,
                               hipStream_t stream)
{
    const long total = (long)P * win * win;
    hipLaunchKernelGGL(hvn_extract_patches_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, img, H, W, coords, win, pad_t, pad_l,
                       out, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
