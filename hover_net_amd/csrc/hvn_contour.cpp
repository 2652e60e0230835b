// hvn_contour.cpp -- host-side border tracing for the per-instance contours of
// /root/reference/models/hovernet/post_proc.py:132-143:
//     cv2.findContours(inst_map_crop, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)[0][0]
// Border following after Suzuki & Abe (1985) with OpenCV 4.3's conventions (imgproc/src/contours.cpp):
//  * the crop is binarised and framed with one background pixel; rows are scanned left to right; an outer border
//    starts at a 0 -> 1 (unvisited) step, a hole border at a (>= 1) -> 0 step;
//  * neighbours are numbered counter-clockwise from east (E, NE, N, NW, W, SW, S, SE); the first neighbour is searched
//    clockwise starting from west (outer) / east (hole), every further one counter-clockwise starting after the pixel
//    we came from; a visited pixel is re-valued with the border's number, negated when the border leaves it to the
//    right, which is what keeps the scan from starting a border twice;
//  * CHAIN_APPROX_SIMPLE keeps a point only where the step direction changes;
//  * RETR_TREE list order: a finished border is linked at the FRONT of its parent's children and the tree is emitted in
//    pre-order, so element [0] is the outer border of the top-level (not inside any hole) piece that was found LAST.
//    An instance of the watershed is 4-connected, i.e. one piece: then [0] simply starts at its first raster pixel.
// Only the candidate for [0] keeps its points; hole borders and nested pieces are followed for their marks only.
// OpenCV is not on the build box.  Pinned by tests/golden/proc_*.npz: the reference's own process() run over an
// independently written python restatement of the same algorithm (oracle/cv2_shim/_suzuki.py).
// O(crop area) per instance, so it stays on the host.
#include <stdint.h>
#include <stdlib.h>

#include <thread>
#include <vector>

#include "../../include/hvn.h"

namespace {

const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct Border {
    bool hole;
    int parent;  // index into the border list, -1 = the image frame
};

struct Tracer {
    std::vector<int32_t> img;  // framed crop: 0 background, 1 unvisited foreground, +-(border index + 2)
    std::vector<Border> borders;
    std::vector<int32_t> cand, cur;  // (x, y) pairs in crop coordinates
    int W2 = 0;

    // follows border `id` from (y0, x0); `keep` collects the CHAIN_APPROX_SIMPLE points into cur
    void follow(int y0, int x0, bool hole, int id, bool keep)
    {
        const int32_t mark = id + 2;
        int32_t *a = img.data();
        auto at = [&](int y, int x) -> int32_t & { return a[y * W2 + x]; };
        if (keep) cur.clear();
        int s = hole ? 0 : 4;
        const int s_first = s;
        int y1, x1;
        do {  // clockwise search for the first neighbour
            s = (s - 1) & 7;
            y1 = y0 + DY[s];
            x1 = x0 + DX[s];
        } while (at(y1, x1) == 0 && s != s_first);
        if (s == s_first) {  // isolated pixel
            at(y0, x0) = -mark;
            if (keep) {
                cur.push_back(x0 - 1);
                cur.push_back(y0 - 1);
            }
            return;
        }
        int y3 = y0, x3 = x0, prev_s = s ^ 4;
        for (;;) {
            const int s_end = s;
            int y4, x4;
            for (;;) {  // counter-clockwise search, starting after the direction we came from
                ++s;
                y4 = y3 + DY[s & 7];
                x4 = x3 + DX[s & 7];
                if (at(y4, x4) != 0) break;
            }
            s &= 7;
            if ((unsigned)(s - 1) < (unsigned)s_end)  // the east neighbour was examined and is background
                at(y3, x3) = -mark;
            else if (at(y3, x3) == 1)
                at(y3, x3) = mark;
            if (s != prev_s) {
                if (keep) {
                    cur.push_back(x3 - 1);
                    cur.push_back(y3 - 1);
                }
                prev_s = s;
            }
            if (y4 == y0 && x4 == x0 && y3 == y1 && x3 == x1) break;
            y3 = y4;
            x3 = x4;
            s = (s + 4) & 7;
        }
    }

    // -> number of points of contours[0], left in cand (crop coordinates)
    long run(const int32_t *inst, int W, int label, int r0, int c0, int h, int w)
    {
        W2 = w + 2;
        img.assign((size_t)(h + 2) * W2, 0);
        for (int y = 0; y < h; ++y) {
            const int32_t *src = inst + (long)(r0 + y) * W + c0;
            int32_t *dst = img.data() + (size_t)(y + 1) * W2 + 1;
            for (int x = 0; x < w; ++x) dst[x] = src[x] == label;
        }
        borders.clear();
        cand.clear();
        for (int y = 1; y <= h; ++y) {
            int32_t *row = img.data() + (size_t)y * W2;
            int lnbd = 0;  // x of the last border pixel met on this row
            int32_t prev = 0;
            for (int x = 1; x <= w; ++x) {
                const int32_t p = row[x];
                if (p == prev) continue;
                bool hole = false;
                if (!(prev == 0 && p == 1)) {
                    if (p != 0 || prev < 1) {  // neither kind of border starts here
                        prev = p;
                        if (prev != 0 && prev != 1) lnbd = x;
                        continue;
                    }
                    if (prev != 1) lnbd = x - 1;
                    hole = true;
                }
                int parent = -1;
                if (lnbd > 0) {
                    const int32_t v = row[lnbd];
                    parent = (v < 0 ? -v : v) - 2;
                    if (borders[parent].hole == hole) parent = borders[parent].parent;
                }
                const int xs = hole ? x - 1 : x;
                lnbd = xs;
                const int id = (int)borders.size();
                borders.push_back(Border{hole, parent});
                const bool top = !hole && parent < 0;
                follow(y, xs, hole, id, top);
                if (top) cand.swap(cur);  // the newest top-level outer border heads the list
                prev = row[x];  // the scan resumes behind the start pixel with its new value
            }
        }
        return (long)cand.size() / 2;
    }
};

}  // namespace

extern "C" HVN_API long hvn_trace_contours(const int32_t *inst, int h, int w, const hvn_inst_rec *recs, int n_rec,
                                           int32_t *pts, long max_pts, int64_t *offs)
{
    if (!inst || !recs || !pts || !offs || h <= 0 || w <= 0 || n_rec < 0) return HVN_E_ARG;
    long crop_px = 0;  // crop pixels to scan
    for (int i = 0; i < n_rec; ++i) {
        const hvn_inst_rec &r = recs[i];
        if (r.area <= 0) continue;
        if (r.rmin < 0 || r.cmin < 0 || r.rmax > h || r.cmax > w || r.rmax <= r.rmin || r.cmax <= r.cmin) return HVN_E_ARG;
        crop_px += (long)(r.rmax - r.rmin) * (r.cmax - r.cmin);
    }
    // Instances are independent: contiguous record ranges go to worker threads (a 2048^2 WSI tile holds ~3000 instances,
    // a 40 000^2 slide ~10^6), each tracing into its own buffer; the offsets are a prefix sum over the per-record counts.
    // (threads only when there is enough to share: an 80 x 80 tile's handful of nuclei is traced faster than a thread starts)
    int n_thr = 1;
    if (n_rec >= 256 && crop_px >= 200000) {
        const char *e = getenv("HVN_HOST_THREADS");
        const unsigned hc = std::thread::hardware_concurrency();
        n_thr = e ? atoi(e) : (int)(hc ? (hc < 16 ? hc : 16) : 4);
        if (n_thr < 1) n_thr = 1;
        if (n_thr > n_rec / 64) n_thr = n_rec / 64;
    }
    std::vector<std::vector<int32_t>> out(n_thr);
    std::vector<int64_t> cnt((size_t)n_rec, 0);
    auto work = [&](int t) {
        Tracer tr;
        std::vector<int32_t> &o = out[t];
        const int lo = (int)((long)n_rec * t / n_thr), hi = (int)((long)n_rec * (t + 1) / n_thr);
        for (int i = lo; i < hi; ++i) {
            const hvn_inst_rec &r = recs[i];
            if (r.area <= 0) continue;
            const long n = tr.run(inst, w, r.label, r.rmin, r.cmin, r.rmax - r.rmin, r.cmax - r.cmin);
            cnt[i] = n;
            for (long k = 0; k < n; ++k) {
                o.push_back(tr.cand[2 * k] + r.cmin);
                o.push_back(tr.cand[2 * k + 1] + r.rmin);
            }
        }
    };
    if (n_thr == 1)
        work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    long total = 0;
    for (int i = 0; i < n_rec; ++i) {
        offs[i] = total;
        total += cnt[i];
    }
    offs[n_rec] = total;
    if (total > max_pts) return HVN_E_SIZE;
    long pos = 0;
    for (int t = 0; t < n_thr; ++t) {
        for (size_t k = 0; k < out[t].size(); ++k) pts[2 * pos + (long)k] = out[t][k];
        pos += (long)out[t].size() / 2;
    }
    return total;
}
