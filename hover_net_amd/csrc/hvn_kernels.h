// Internal launch interfaces shared by the .hip translation units of libhvn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ConvArgs {
    const float *x;
    long xsn, xsy, xsx;
    int H, W, Cin;
    const float *w;
    const float *bias, *pre_s, *pre_b, *post_s, *post_b;
    const float *res;
    long rsn, rsy, rsx;
    float *y;
    long ysn, ysy, ysx;
    int N, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, relu;
    long M;
    long m_tiles;
    int n_tiles;
    int groups;
    const float *x2;      // optional second 1x1 input (channels appended to the reduction)
    long x2sn, x2sy, x2sx;
    int Cin2, stride2;
    int nbatch;           // > 1: blockIdx.y selects one of nbatch independent problems
    long xb, wb, yb;      // element strides between them
};

int hvn_launch_conv(const ConvArgs &a, int tile_n, hipStream_t stream);

struct Conv0Args {
    const void *img;      // uint8 or float32
    long isn, isy, isx, isc;  // strides in elements
    int is_f32;
    int H, W;             // input extent
    const float *w;       // [7][7][3][64]
    const float *bias;    // [64]
    float *y;
    long ysn, ysy, ysx;
    int N, Ho, Wo, pad;
};
int hvn_launch_conv0(const Conv0Args &a, hipStream_t stream);

struct UpAddArgs {
    const float *lo;
    long lsn, lsy, lsx;
    const float *skip;
    long ssn, ssy, ssx;
    float *y;
    long ysn, ysy, ysx;
    int N, H, W, C;  // output extent
};
int hvn_launch_upadd(const UpAddArgs &a, hipStream_t stream);

struct HeadArgs {
    const float *x;
    long xsn, xsy, xsx;
    const float *w;     // [cout][64]
    const float *bias;  // [cout]
    float *y;           // NCHW [N][cout][H][W]
    int N, H, W, Cout;
};
int hvn_launch_head(const HeadArgs &a, hipStream_t stream);

struct PredMapArgs {
    const float *tp, *np, *hv;  // NCHW logits
    float *y;                   // [N][H][W][3|4]
    int N, H, W, nr_types;      // nr_types = 0: no type channel
};
int hvn_launch_predmap(const PredMapArgs &a, hipStream_t stream);

struct WinoArgs {
    const float *x;       // WINO_IN: input view; WINO_OUT: M [n*n][tiles][C] per sample (n = m + 4)
    long xsn, xsy, xsx;
    float *y;             // WINO_IN: V [n*n][tiles][C] per sample; WINO_OUT: output view
    long ysn, ysy, ysx;
    const float *mat;     // B^T (n x n) or A^T (m x n)
    const float *bias;    // WINO_OUT only
    int N, H, W, C;       // WINO_IN: input window extent / channels; WINO_OUT: output extent / cout
    int ty, tx, pad, relu;
    int m;                // output tile edge: 2 or 4
};
int hvn_launch_wino_in(const WinoArgs &a, hipStream_t stream);
int hvn_launch_wino_out(const WinoArgs &a, hipStream_t stream);
