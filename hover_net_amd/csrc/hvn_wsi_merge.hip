// hvn_wsi_merge.hip -- the sequential instance-map merge of whole-slide inference on a DEVICE-resident map.
//
// Reference: /root/reference/infer/wsi.py:569-599 (post_proc_normal_tile_callback) and :602-677
// (post_proc_fixing_tile_callback) with _remove_inst (:51-60).  The callbacks "must be in sequential ordering": every tile's id
// offset is the running maximum id and the fix-up windows overlap.  That order is kept -- one tile at a time, in tile order --
// but each tile's array work (np.unique over half a megapixel, np.isin, the window rewrite: ~5 ms of numpy per tile, 7.3 of the
// 13.85 s of stage 2 on a 40 000^2 slide) becomes four launches over the window, and the 6.4 GB int32 map never leaves HBM
// until the slide is done.  The dictionary side (which ids die, which new ids enter) goes back to the host as two short lists.
//
// fixing tile, in the reference's words:
//   roi        = wsi_inst_map[window]
//   on_edge    = np.unique(edge pixels of roi)[1:]        -- "[1:] exclude background": drops the SMALLEST value, whatever it is
//   inner      = np.unique(roi)[1:] minus on_edge         -- old instances wholly inside the window: removed from map and dict
//   touching   = new ids that overlap a remaining (kept, split) old instance: dropped
//   new_inner  = the other new ids: written with + offset; entered into the dict if they have a contour and are in np.unique(pred_inst)[1:]
//                (again "[1:]": a tile without background loses its smallest id from the DICT, not from the map)
// Ids are flagged in epoch-stamped tables (no clearing between tiles).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hvn.h"

#define MG_T 256

struct MergeArgs {
    int32_t *map;
    long map_w;
    int y0, x0, h, w;
    const int32_t *pred;      // [h][w] local ids
    int32_t off, epoch;
    int32_t *edge_flag, *rem_flag;   // [cap]
    long cap;
    int32_t *removed;
    int removed_cap;
    int32_t *counters;        // 0: removed ids, 1: smallest value on the window edge, 2: smallest value in the window, 3: smallest value of the new tile,
                              // 4: window pixels whose id lies beyond the id tables (>= cap): the caller sized the tables too small -> it raises
    uint8_t *touching;        // [n_local + 1]
    int n_local;
};

__global__ __launch_bounds__(MG_T) void mg_normal(int32_t *map, long map_w, int y0, int x0, int h, int w, const int32_t *pred, int32_t off)
{
    const long i = (long)blockIdx.x * MG_T + threadIdx.x;
    if (i >= (long)h * w) return;
    const int y = (int)(i / w), x = (int)(i - (long)y * w);
    const int32_t p = pred[i];
    map[(long)(y0 + y) * map_w + (x0 + x)] = p > 0 ? p + off : 0;     // pred_inst[pred_inst > 0] += offset; window = pred_inst
}

__global__ __launch_bounds__(MG_T) void mg_init(MergeArgs a)
{
    const int i = blockIdx.x * MG_T + threadIdx.x;
    if (i == 0) {
        a.counters[0] = 0;
        a.counters[1] = 0x7fffffff;
        a.counters[2] = 0x7fffffff;
        a.counters[3] = 0x7fffffff;
        a.counters[4] = 0;
    }
    if (i <= a.n_local) a.touching[i] = 0;
}

__global__ __launch_bounds__(MG_T) void mg_scan(MergeArgs a)
{
    const long i = (long)blockIdx.x * MG_T + threadIdx.x;
    int32_t id = 0x7fffffff, eid = 0x7fffffff, pid = 0x7fffffff;
    if (i < (long)a.h * a.w) {
        const int y = (int)(i / a.w), x = (int)(i - (long)y * a.w);
        id = a.map[(long)(a.y0 + y) * a.map_w + (a.x0 + x)];
        pid = a.pred[i];
        if (y == 0 || y == a.h - 1 || x == 0 || x == a.w - 1) {
            eid = id;
            if (id > 0 && id < a.cap) a.edge_flag[id] = a.epoch;
        }
    }
    // block minimum, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        id = min(id, __shfl_xor(id, o));
        eid = min(eid, __shfl_xor(eid, o));
        pid = min(pid, __shfl_xor(pid, o));
    }
    if ((threadIdx.x & 63) == 0) {
        if (pid != 0x7fffffff) atomicMin(&a.counters[3], pid);
        if (id != 0x7fffffff) atomicMin(&a.counters[2], id);
        if (eid != 0x7fffffff) atomicMin(&a.counters[1], eid);
    }
}

__global__ __launch_bounds__(MG_T) void mg_remove(MergeArgs a)
{
    const long i = (long)blockIdx.x * MG_T + threadIdx.x;
    if (i >= (long)a.h * a.w) return;
    const int y = (int)(i / a.w), x = (int)(i - (long)y * a.w);
    int32_t *m = a.map + (long)(a.y0 + y) * a.map_w + (a.x0 + x);
    const int32_t id = *m;
    if (id >= a.cap) atomicAdd(&a.counters[4], 1);      // never silently: the host raises on a non-zero count
    if (id <= 0 || id >= a.cap) return;
    const int32_t edge_min = a.counters[1], roi_min = a.counters[2];
    // np.unique(...)[1:] drops the smallest value: the background 0 if the window (edge) holds any, else the smallest id
    const bool on_edge = a.edge_flag[id] == a.epoch && !(edge_min > 0 && id == edge_min);
    const bool smallest = roi_min > 0 && id == roi_min;
    if (on_edge || smallest) return;
    *m = 0;
    if (atomicExch(&a.rem_flag[id], a.epoch) != a.epoch) {
        const int k = atomicAdd(&a.counters[0], 1);
        if (k < a.removed_cap) a.removed[k] = id;
    }
}

__global__ __launch_bounds__(MG_T) void mg_touch(MergeArgs a)
{
    const long i = (long)blockIdx.x * MG_T + threadIdx.x;
    if (i >= (long)a.h * a.w) return;
    const int32_t p = a.pred[i];
    if (p <= 0 || p > a.n_local) return;
    const int y = (int)(i / a.w), x = (int)(i - (long)y * a.w);
    if (a.map[(long)(a.y0 + y) * a.map_w + (a.x0 + x)] > 0) a.touching[p] = 1;
}

__global__ __launch_bounds__(MG_T) void mg_write(MergeArgs a)
{
    const long i = (long)blockIdx.x * MG_T + threadIdx.x;
    if (i >= (long)a.h * a.w) return;
    const int32_t p = a.pred[i];
    if (p <= 0 || p > a.n_local || a.touching[p]) return;
    const int y = (int)(i / a.w), x = (int)(i - (long)y * a.w);
    a.map[(long)(a.y0 + y) * a.map_w + (a.x0 + x)] += p + a.off;       // window = roi + pred_inst (roi is 0 under a non-touching new id)
}

extern "C" {

int hvn_wsi_merge_normal(int32_t *inst_map, int64_t map_w, int y0, int x0, int h, int w, const int32_t *pred_inst, int32_t off, void *stream)
{
    if (!inst_map || !pred_inst || h <= 0 || w <= 0 || map_w < x0 + w || y0 < 0 || x0 < 0) return HVN_E_ARG;
    const long n = (long)h * w;
    hipLaunchKernelGGL(mg_normal, dim3((unsigned)((n + MG_T - 1) / MG_T)), dim3(MG_T), 0, (hipStream_t)stream, inst_map, (long)map_w, y0, x0, h, w, pred_inst, off);
    return hipGetLastError() == hipSuccess ? HVN_OK : HVN_E_LAUNCH;
}

int hvn_wsi_merge_fixing(int32_t *inst_map, int64_t map_w, int y0, int x0, int h, int w, const int32_t *pred_inst, int32_t n_local,
                         int32_t off, int32_t epoch, int32_t *id_flags, int64_t cap, int32_t *removed, int32_t removed_cap,
                         int32_t *counters, uint8_t *touching, void *stream)
{
    if (!inst_map || !pred_inst || !id_flags || !removed || !counters || !touching || h <= 0 || w <= 0 || map_w < x0 + w || y0 < 0 || x0 < 0 ||
        n_local < 0 || cap <= 0 || removed_cap <= 0 || epoch <= 0)
        return HVN_E_ARG;
    MergeArgs a;
    a.map = inst_map; a.map_w = (long)map_w; a.y0 = y0; a.x0 = x0; a.h = h; a.w = w; a.pred = pred_inst; a.off = off; a.epoch = epoch;
    a.edge_flag = id_flags; a.rem_flag = id_flags + cap; a.cap = (long)cap; a.removed = removed; a.removed_cap = removed_cap;
    a.counters = counters; a.touching = touching; a.n_local = n_local;
    const hipStream_t s = (hipStream_t)stream;
    const long n = (long)h * w;
    const dim3 grid((unsigned)((n + MG_T - 1) / MG_T)), blk(MG_T);
    hipLaunchKernelGGL(mg_init, dim3((unsigned)((n_local + 1 + MG_T - 1) / MG_T)), blk, 0, s, a);
    hipLaunchKernelGGL(mg_scan, grid, blk, 0, s, a);
    hipLaunchKernelGGL(mg_remove, grid, blk, 0, s, a);
    hipLaunchKernelGGL(mg_touch, grid, blk, 0, s, a);
    hipLaunchKernelGGL(mg_write, grid, blk, 0, s, a);
    return hipGetLastError() == hipSuccess ? HVN_OK : HVN_E_LAUNCH;
}

}  // extern "C"
