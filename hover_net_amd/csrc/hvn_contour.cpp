// hvn_contour.cpp -- host-side outer-border tracing for the per-instance contours of
// /root/reference/models/hovernet/post_proc.py:132-143:
//     cv2.findContours(inst_map_crop, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)[0][0]
// Border following after Suzuki & Abe (1985) as OpenCV 4.3 implements it (imgproc/src/contours.cpp,
// icvFetchContour): 8-connected foreground, the outer border starts at the first foreground pixel in
// raster order, neighbours are numbered counter-clockwise from east (E, NE, N, NW, W, SW, S, SE), the first
// neighbour is searched clockwise starting from west, every further one counter-clockwise starting after
// the direction we came from; CHAIN_APPROX_SIMPLE keeps a point only where the step direction changes.
// OpenCV itself is not available on the build box: the point ORDER is restated from memory
// (PARITY UNPINNED); the SET of border pixels is checked against an independent definition in the tests.
// O(perimeter) per instance over its bounding-box crop, so it stays on the host.
#include <stdint.h>

#include <vector>

#include "../../include/hvn.h"

namespace {

struct Crop {
    const int32_t *inst;
    int W, label, r0, c0, h, w;  // crop origin / extent inside the full map
    // foreground test with the implicit 1-px zero border OpenCV adds
    inline bool fg(int y, int x) const
    {
        return (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w && inst[(long)(r0 + y) * W + (c0 + x)] == label;
    }
};

const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

// appends (x, y) points in full-map coordinates; returns the number of points written
long trace_outer(const Crop &c, std::vector<int32_t> &out)
{
    // start: first foreground pixel in raster order (its west neighbour is background by construction)
    int y0 = -1, x0 = -1;
    for (int y = 0; y < c.h && y0 < 0; ++y)
        for (int x = 0; x < c.w; ++x)
            if (c.fg(y, x)) {
                y0 = y;
                x0 = x;
                break;
            }
    if (y0 < 0) return 0;
    const long first = (long)out.size() / 2;
    auto emit = [&](int y, int x) {
        out.push_back(c.c0 + x);
        out.push_back(c.r0 + y);
    };
    // clockwise search for the first neighbour, starting from west (s = 4)
    int s = 4;
    const int s_end = 4;
    int y1, x1;
    do {
        s = (s - 1) & 7;
        y1 = y0 + DY[s];
        x1 = x0 + DX[s];
    } while (!c.fg(y1, x1) && s != s_end);
    if (s == s_end) {  // isolated pixel
        emit(y0, x0);
        return 1;
    }
    int y3 = y0, x3 = x0;
    int prev_s = s ^ 4;
    for (;;) {
        int y4, x4;
        for (;;) {  // counter-clockwise search, starting after the direction we came from
            s = (s + 1) & 7;
            y4 = y3 + DY[s];
            x4 = x3 + DX[s];
            if (c.fg(y4, x4)) break;
        }
        if (s != prev_s) emit(y3, x3);  // CHAIN_APPROX_SIMPLE: keep the corners only
        prev_s = s;
        const bool done = (y4 == y0 && x4 == x0 && y3 == y1 && x3 == x1);
        if (done) break;
        y3 = y4;
        x3 = x4;
        s = (s + 4) & 7;
    }
    return (long)out.size() / 2 - first;
}

}  // namespace

extern "C" HVN_API long hvn_trace_contours(const int32_t *inst, int h, int w, const hvn_inst_rec *recs, int n_rec,
                                           int32_t *pts, long max_pts, int64_t *offs)
{
    if (!inst || !recs || !pts || !offs || h <= 0 || w <= 0 || n_rec < 0) return HVN_E_ARG;
    std::vector<int32_t> out;
    out.reserve(1024);
    long total = 0;
    for (int i = 0; i < n_rec; ++i) {
        offs[i] = total;
        const hvn_inst_rec &r = recs[i];
        if (r.area <= 0) continue;
        if (r.rmin < 0 || r.cmin < 0 || r.rmax > h || r.cmax > w || r.rmax <= r.rmin || r.cmax <= r.cmin) return HVN_E_ARG;
        Crop c{inst, w, r.label, r.rmin, r.cmin, r.rmax - r.rmin, r.cmax - r.cmin};
        out.clear();
        const long n = trace_outer(c, out);
        if (total + n > max_pts) return HVN_E_SIZE;
        for (long k = 0; k < 2 * n; ++k) pts[2 * total + k] = out[k];
        total += n;
    }
    offs[n_rec] = total;
    return total;
}
