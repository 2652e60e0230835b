"""Output writers of the inference managers that sit right behind the hot path -- host glue kept format-compatible:
`save_json` writes the `{"mag": ..., "nuc": {id: {bbox, centroid, contour, type_prob, type}}}` protocol of
`infer/base.py:80-94` (what QuPath import and `compute_stats.py` read)."""
import json

import numpy as np


def save_json(path, inst_info, mag=None):
    """inst_info: the dict `post_proc.process` / `WsiInference.run` return.  Returns the JSON-able dict."""
    new = {}
    for inst_id, info in inst_info.items():
        new[int(inst_id)] = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in info.items()}
    with open(path, "w") as handle:
        json.dump({"mag": mag, "nuc": new}, handle)
    return new
