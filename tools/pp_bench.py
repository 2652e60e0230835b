#!/usr/bin/env python
"""Times the on-GPU instance separation alone on the bench's structured synthetic maps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hover_net_amd.post_proc import PostProc
from hover_net_amd.synth import synth_pred_maps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 80
klo = int(sys.argv[3]) if len(sys.argv) > 3 else 5
khi = int(sys.argv[4]) if len(sys.argv) > 4 else 40
mode = sys.argv[5] if len(sys.argv) > 5 else "struct"      # struct | noise (tile-filling smooth-noise blobs, what a random-init net emits) | quant
if mode == "noise":
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(7))

    def smooth(a, it=3):
        for _ in range(it):
            a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5.0
        return a

    f = np.stack([smooth(rng.normal(0, 1, (n, hw, hw)).transpose(1, 2, 0)).transpose(2, 0, 1) for _ in range(4)], -1)
    f = f / f.std()
    f[..., 1] = 0.75 + 0.5 * f[..., 1]
    pred = torch.from_numpy(f.astype(np.float32)).cuda()
else:
    arr = synth_pred_maps(n, hw, hw, 5, seed=100, k_lo=klo, k_hi=khi, noise=0.0 if mode == "quant" else 0.02)[0]
    if mode == "quant":
        import numpy as np
        arr[..., 2:] = np.round(arr[..., 2:] * 4) / 4
    pred = torch.from_numpy(arr).cuda()
pp = PostProc("cuda")
for _ in range(2):
    inst = pp.separate(pred)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    inst = pp.separate(pred)
torch.cuda.synchronize()
print("separate[%s]: %.3f ms per batch of %d %dx%d maps, %d instances" % (mode, (time.perf_counter() - t) / 5 * 1e3, n, hw, hw,
      sum(len(torch.unique(i)) - 1 for i in inst)))
