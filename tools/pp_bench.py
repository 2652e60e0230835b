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
pred = torch.from_numpy(synth_pred_maps(n, hw, hw, 5, seed=100, k_lo=klo, k_hi=khi)[0]).cuda()
pp = PostProc("cuda")
for _ in range(2):
    inst = pp.separate(pred)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    inst = pp.separate(pred)
torch.cuda.synchronize()
print("separate: %.3f ms per batch of %d %dx%d maps, %d instances" % ((time.perf_counter() - t) / 5 * 1e3, n, hw, hw,
      sum(len(torch.unique(i)) - 1 for i in inst)))
