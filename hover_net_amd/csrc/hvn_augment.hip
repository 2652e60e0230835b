// hvn_augment.hip -- training-time augmentation on the GPU for a resident patch set.
//
// Stands behind dataloader/train_loader.py:76-199 (`FileLoader.__getitem__` + `__get_augmentation`) and the explicit image
// functions of dataloader/augs.py:36-113 that its `iaa.Lambda`s call.  The reference runs this per sample on 16 CPU
// workers; here the extracted training patches live in HBM (a 540 x 540 x (3 u8 + 2 i32) patch is 3.2 MB: ~90 000 of them
// fit beside the model) and one batch costs two launches:
//   hvn_aug_shape_k   affine warp (nearest, constant 0) + centre crop + flips, fused into ONE gather per output pixel for the
//                     image and the annotation planes, reading sample `src` of the resident set;
//   hvn_aug_input_k   one of {Gaussian blur, median blur, additive noise}, then hue / saturation / brightness / contrast in
//                     the sampled order, per pixel on the cropped image.
// Both are byte movers on < 1 MB per sample (HBM / L2 bound; no MFMA).  The parameters of every sample are drawn on the host
// (`hover_net_amd/augment.py`) and arrive as one `hvn_aug_sample` record per output sample.
// Arithmetic mirrors oracle/augment_np.py operation for operation (float64 where numpy promotes to float64, float32 in the
// HSV -> RGB step, integer tables for RGB -> HSV / grey), so the parity tests are bit-exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hvn.h"
#include "hvn_kernels.h"

#define AUG_T 256

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AUG_T) void hvn_aug_shape_k(const uint8_t *img, const int32_t *ann, int H, int W, int C, const hvn_aug_sample *prm,
                                                          int oh, int ow, int y0, int x0, uint8_t *oimg, int32_t *oann, long total)
{
    const long i = (long)blockIdx.x * AUG_T + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % ow);
    const long t = i / ow;
    const int y = (int)(t % oh);
    const int n = (int)(t / oh);
    const hvn_aug_sample &s = prm[n];
    const int xs = s.flip_lr ? ow - 1 - x : x;
    const int ys = s.flip_ud ? oh - 1 - y : y;
    const double xd = (double)(xs + x0), yd = (double)(ys + y0);
    const double sxf = s.inv[0] * xd + s.inv[1] * yd + s.inv[2];
    const double syf = s.inv[3] * xd + s.inv[4] * yd + s.inv[5];
    const double fx = floor(sxf + 0.5), fy = floor(syf + 0.5);
    const bool ok = fx >= 0.0 && fx < (double)W && fy >= 0.0 && fy < (double)H;
    uint8_t *di = oimg + i * 3;
    int32_t *da = oann + i * C;
    if (ok) {
        const long src = ((long)s.src * H + (long)fy) * W + (long)fx;
        const uint8_t *si = img + src * 3;
        di[0] = si[0];
        di[1] = si[1];
        di[2] = si[2];
        for (int c = 0; c < C; ++c) da[c] = ann[src * C + c];
    } else {
        di[0] = di[1] = di[2] = 0;
        for (int c = 0; c < C; ++c) da[c] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ inline int aug_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cv2.cvtColor(uint8, RGB2HSV): integer path with rounded reciprocal tables, H in 0..179
__device__ inline void aug_rgb2hsv(const int *p, int *hsv)
{
    const int r = p[0], g = p[1], b = p[2];
    const int v = max(max(r, g), b), vmin = min(min(r, g), b);
    const int diff = v - vmin;
    const long sdiv = v ? (long)rint((double)(255 << 12) / (1.0 * (double)v)) : 0;
    const long hdiv = diff ? (long)rint((double)(180 << 12) / (6.0 * (double)diff)) : 0;
    const long sat = ((long)diff * sdiv + (1 << 11)) >> 12;
    long hh = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
    hh = (hh * hdiv + (1 << 11)) >> 12;
    if (hh < 0) hh += 180;
    hsv[0] = (int)(hh & 255);
    hsv[1] = (int)(sat & 255);
    hsv[2] = v;
}

// cv2.cvtColor(uint8, HSV2RGB): float32 sector arithmetic, saturate_cast<uchar>(x * 255)
__device__ inline void aug_hsv2rgb(const int *hsv, int *p)
{
    float h = (float)hsv[0] * (float)(6.0 / 180.0);
    const float s = (float)hsv[1] * (float)(1.0 / 255.0);
    const float v = (float)hsv[2] * (float)(1.0 / 255.0);
    float r, g, b;
    if (hsv[1] == 0) {
        r = g = b = v;
    } else {
        if (h >= 6.f) h = h - 6.f;
        int sector = (int)floorf(h);
        float hf = h - (float)sector;
        if (sector < 0 || sector >= 6) {
            sector = 0;
            hf = 0.f;
        }
        float tab[4];
        tab[0] = v;
        tab[1] = v * (1.f - s);
        tab[2] = v * (1.f - s * hf);
        tab[3] = v * (1.f - s * (1.f - hf));
        const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
        b = tab[sd[sector][0]];
        g = tab[sd[sector][1]];
        r = tab[sd[sector][2]];
    }
    p[0] = aug_clampi((int)rintf(r * 255.f), 0, 255);
    p[1] = aug_clampi((int)rintf(g * 255.f), 0, 255);
    p[2] = aug_clampi((int)rintf(b * 255.f), 0, 255);
}

__device__ inline int aug_clip_trunc(double v) { return v < 0.0 ? 0 : (v > 255.0 ? 255 : (int)v); }

__global__ __launch_bounds__(AUG_T) void hvn_aug_input_k(const uint8_t *src, const hvn_aug_sample *prm, const float *noise, int H, int W, uint8_t *dst,
                                                          long total)
{
    const long i = (long)blockIdx.x * AUG_T + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W);
    const long t = i / W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const hvn_aug_sample &s = prm[n];
    const uint8_t *im = src + (long)n * H * W * 3;
    int p[3];
    if (s.kind == 0) {  // cv2.GaussianBlur(ksize in {1,3,5}^2, sigma 0, BORDER_REPLICATE): exact rational, round half up
        const int kx = s.p0, ky = s.p1, rx = kx / 2, ry = ky / 2;
        const int t3[3] = {1, 2, 1}, t5[5] = {1, 4, 6, 4, 1}, t1[1] = {1};
        const int *tx = kx == 5 ? t5 : (kx == 3 ? t3 : t1), *ty = ky == 5 ? t5 : (ky == 3 ? t3 : t1);
        const long den = (long)(kx == 5 ? 16 : (kx == 3 ? 4 : 1)) * (ky == 5 ? 16 : (ky == 3 ? 4 : 1));
        long acc[3] = {0, 0, 0};
        for (int j = 0; j < ky; ++j) {
            const int yy = aug_clampi(y + j - ry, 0, H - 1);
            for (int k = 0; k < kx; ++k) {
                const int xx = aug_clampi(x + k - rx, 0, W - 1);
                const uint8_t *q = im + ((long)yy * W + xx) * 3;
                const long wgt = (long)ty[j] * tx[k];
                acc[0] += wgt * q[0];
                acc[1] += wgt * q[1];
                acc[2] += wgt * q[2];
            }
        }
        for (int c = 0; c < 3; ++c) p[c] = (int)((2 * acc[c] + den) / (2 * den));
    } else if (s.kind == 1 && s.p0 > 1) {  // cv2.medianBlur(k in {3,5}), per channel, BORDER_REPLICATE
        const int k = s.p0, r = k / 2, cnt = k * k;
        for (int c = 0; c < 3; ++c) {
            uint8_t v[25];
            int m = 0;
            for (int j = 0; j < k; ++j) {
                const int yy = aug_clampi(y + j - r, 0, H - 1);
                for (int kk = 0; kk < k; ++kk) {
                    const int xx = aug_clampi(x + kk - r, 0, W - 1);
                    const uint8_t val = im[((long)yy * W + xx) * 3 + c];
                    int q = m++;
                    while (q > 0 && v[q - 1] > val) {  // insertion sort (<= 25 values)
                        v[q] = v[q - 1];
                        --q;
                    }
                    v[q] = val;
                }
            }
            p[c] = v[cnt / 2];
        }
    } else {
        const uint8_t *q = im + ((long)y * W + x) * 3;
        p[0] = q[0];
        p[1] = q[1];
        p[2] = q[2];
        if (s.kind == 2) {  // additive Gaussian noise: samples rounded to integers, added, saturated
            const float *z = noise + i * 3;
            for (int c = 0; c < 3; ++c) {
                const float nz = z[s.per_channel ? c : 0] * s.noise_scale;
                p[c] = aug_clampi(p[c] + (int)rintf(nz), 0, 255);
            }
        }
    }
    for (int k = 0; k < 4; ++k) {
        const int op = s.order[k];
        if (op == 0) {  // augs.py:62-75 add_to_hue
            int hsv[3];
            aug_rgb2hsv(p, hsv);
            double h = fmod((double)hsv[0] + s.hue, 180.0);
            if (h != 0.0 && h < 0.0) h += 180.0;   // numpy's float mod: result takes the sign of the divisor
            hsv[0] = (int)h & 255;                // float64 -> uint8 store: truncation
            aug_hsv2rgb(hsv, p);
        } else if (op == 1) {  // augs.py:79-87 add_to_saturation (s.sat = 1 + draw)
            const long grey = ((long)p[0] * 4899 + (long)p[1] * 9617 + (long)p[2] * 1868 + (1 << 13)) >> 14;
            const double gv = (double)grey * (1.0 - s.sat);
            for (int c = 0; c < 3; ++c) p[c] = aug_clip_trunc((double)p[c] * s.sat + gv);
        } else if (op == 2) {  // augs.py:103-109 add_to_brightness
            for (int c = 0; c < 3; ++c) p[c] = aug_clip_trunc((double)p[c] + s.bright);
        }
        // op == 3: add_to_contrast returns its input (augs.py:96-97 clips `img`, not the adjusted array); op < 0: skipped
    }
    uint8_t *d = dst + i * 3;
    d[0] = (uint8_t)p[0];
    d[1] = (uint8_t)p[1];
    d[2] = (uint8_t)p[2];
}

// ---------------------------------------------------------------------------------------------------------------------
int hvn_launch_aug_shape(const uint8_t *img, const int32_t *ann, int h, int w, int c, const hvn_aug_sample *prm, int n, int oh, int ow, uint8_t *oimg,
                         int32_t *oann, hipStream_t stream)
{
    const long total = (long)n * oh * ow;
    const int y0 = (int)((h - oh) * 0.5), x0 = (int)((w - ow) * 0.5);   // cropping_center / CropToFixedSize(position="center")
    hipLaunchKernelGGL(hvn_aug_shape_k, dim3((unsigned)((total + AUG_T - 1) / AUG_T)), dim3(AUG_T), 0, stream, img, ann, h, w, c, prm, oh, ow, y0, x0,
                       oimg, oann, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hvn_launch_aug_input(const uint8_t *src, const hvn_aug_sample *prm, const float *noise, int n, int h, int w, uint8_t *dst, hipStream_t stream)
{
    const long total = (long)n * h * w;
    hipLaunchKernelGGL(hvn_aug_input_k, dim3((unsigned)((total + AUG_T - 1) / AUG_T)), dim3(AUG_T), 0, stream, src, prm, noise, h, w, dst, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
