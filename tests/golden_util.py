"""Readers for the flat-array fixtures under tests/golden/."""
import numpy as np


def golden_dicts(z):
    """proc_*.npz -> one inst_info_dict per map, in the reference's key order."""
    out = []
    mo, co = z["map_off"], z["contour_off"]
    for m in range(len(mo) - 1):
        d = {}
        for j in range(mo[m], mo[m + 1]):
            t = int(z["type"][j])
            d[int(z["ids"][j])] = {"bbox": z["bbox"][j], "centroid": z["centroid"][j], "contour": z["contour_pts"][co[j]:co[j + 1]],
                                  "type": None if t < 0 else t, "type_prob": None if np.isnan(z["type_prob"][j]) else float(z["type_prob"][j])}
        out.append(d)
    return out


def assert_same_info(got, want):
    assert list(got.keys()) == list(want.keys())
    for k, w in want.items():
        g = got[k]
        assert np.asarray(g["bbox"]).tolist() == w["bbox"].tolist(), k
        assert np.asarray(g["centroid"]).tolist() == w["centroid"].tolist(), k
        assert g["contour"].dtype == np.int32 and g["contour"].tolist() == w["contour"].tolist(), k
        assert g["type"] == w["type"] and g["type_prob"] == w["type_prob"], k
