"""CPU: tests/pq_util.pq (the metric behind the declared bf16 tolerance) against the reference's own `get_fast_pq` + `remap_label`
(metrics/stats_utils.py), imported unmodified in the build container."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pq_util import pq

_REF = r'''
import sys, types, json
sys.path.insert(0, "/root/reference")
sys.modules["cv2"] = types.ModuleType("cv2")
import numpy as np
from metrics.stats_utils import get_fast_pq, remap_label          # the reference, unmodified
d = np.load(sys.argv[1])
out = []
for t, p in zip(d["true"], d["pred"]):
    if t.max() == 0 or p.max() == 0:
        out.append(None)                                           # the reference's function does not handle empty maps
        continue
    out.append([float(v) for v in get_fast_pq(remap_label(t), remap_label(p))[0]])
print("PQ " + json.dumps(out))
'''


def _maps(seed, n=10, h=64):
    rng = np.random.default_rng(seed)
    true, pred = [], []
    for i in range(n):
        t = np.zeros((h, h), np.int32)
        for k in range(1, int(rng.integers(2, 12))):
            y, x, a, b = rng.integers(0, h - 12), rng.integers(0, h - 12), rng.integers(4, 12), rng.integers(4, 12)
            t[y:y + a, x:x + b] = k * 3                           # non-contiguous ids
        p = np.roll(t, (int(rng.integers(-2, 3)), int(rng.integers(-2, 3))), (0, 1)).copy()
        ids = np.unique(p)[1:]
        if len(ids) > 2:
            p[p == ids[0]] = 0                                    # a missed instance
            p[p == ids[1]] = ids[2]                               # a merge
        p[:5, :5] = 999                                           # a spurious one
        true.append(t)
        pred.append(p)
    return np.stack(true), np.stack(pred)


@pytest.mark.skipif(not os.path.exists("/root/reference/metrics/stats_utils.py"), reason="needs the reference tree (build container only)")
def test_pq_equals_the_references_get_fast_pq(tmp_path):
    true, pred = _maps(1)
    np.savez(tmp_path / "m.npz", true=true, pred=pred)
    r = subprocess.run([sys.executable, "-c", _REF, str(tmp_path / "m.npz")], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg"))
    assert r.returncode == 0 and "PQ " in r.stdout, (r.stdout[-500:], r.stderr[-2500:])
    ref = json.loads(r.stdout.split("PQ ", 1)[1].splitlines()[0])
    checked = 0
    for t, p, w in zip(true, pred, ref):
        if w is None:
            continue
        assert abs(pq(t, p) - w[2]) < 1e-9, (pq(t, p), w)       # [dq, sq, dq * sq]
        assert abs(pq(t, t) - 1.0) < 2e-6
        checked += 1
    assert checked >= 8


def test_pq_edge_cases():
    z = np.zeros((8, 8), np.int32)
    one = z.copy()
    one[2:5, 2:5] = 4
    assert pq(z, z) == 1.0 and pq(one, z) == 0.0 and pq(z, one) == 0.0 and abs(pq(one, one) - 1.0) < 2e-6
