"""Panoptic quality of one predicted instance map against one reference map, restated from the reference's metric
(metrics/stats_utils.py:178-260 `get_fast_pq` after :360 `remap_label`): unique IoU > 0.5 pairing, DQ = TP / (TP + FP/2 + FN/2),
SQ = sum of paired IoUs / (TP + 1e-6), PQ = DQ x SQ; two empty maps score 1.  Pinned to the reference's own function in
tests/test_oracle_metrics.py; used by the bf16 tolerance test on the GPU box, where the reference tree does not exist."""
import numpy as np


def pq(true, pred):
    tl, pl = np.unique(true)[1:] if (true == 0).any() else np.unique(true), np.unique(pred)[1:] if (pred == 0).any() else np.unique(pred)
    tl, pl = tl[tl != 0], pl[pl != 0]
    if len(tl) == 0 and len(pl) == 0:
        return 1.0
    tp, iou_sum, used = 0, 0.0, set()
    for t in tl:
        m = true == t
        cand, cnt = np.unique(pred[m], return_counts=True)
        for c, k in zip(cand, cnt):
            if c == 0 or c in used:
                continue
            iou = k / float(m.sum() + (pred == c).sum() - k)
            if iou > 0.5:          # IoU > 0.5 pairs are unique by construction
                tp += 1
                iou_sum += iou
                used.add(c)
                break
    fp, fn = len(pl) - tp, len(tl) - tp
    return (tp / (tp + 0.5 * fp + 0.5 * fn)) * (iou_sum / (tp + 1e-6))
