#!/usr/bin/env python
"""Round 6: the conv kernels' epilogues exist in 8 operand-set forms (bias + ReLU / residual / block BN-ReLU present or not) instead of one that
computes absent operands as identities.  This check runs the whole network -- fp32 'original' 5 types and bf16 'fast' 6 types, batch 3 -- once with
the shipped library and once with the A/B build that keeps the ONE full epilogue of rounds 1-5 everywhere (lib.VARIANTS["fullepi"]), each in
its own process, and compares logits and prediction maps bit for bit.
usage: python tools/epilogue_forms_check.py            (needs `lib.build_variant("fullepi")` next to the default library)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from hover_net_amd import net_desc, run_desc
from hover_net_amd.synth import synth_state_dict, synth_tiles
out = {}
for mode, nt, size, dt in (("original", 5, 270, "fp32"), ("fast", 6, 256, "bf16"), ("fast", 6, 256, "fp32")):
    net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3)
    net.load_state_dict(synth_state_dict(mode, nt, seed=7), strict=True)
    net.compute_dtype = dt
    net = net.cuda().eval()
    tiles = torch.from_numpy(synth_tiles(3, size, seed=8))
    pred = run_desc.infer_step_device(tiles, net)
    eng = net.engine(3)
    for k in eng.logits:
        out["%%s_%%s_%%s" %% (mode, dt, k)] = eng.logits[k][:3].cpu().numpy()
    out["%%s_%%s_pred" %% (mode, dt)] = pred.cpu().numpy()
np.savez(sys.argv[1], **out)
'''


def main():
    tmp = tempfile.mkdtemp(prefix="hvn_epi_")
    files = {}
    for variant in ("", "fullepi"):
        f = os.path.join(tmp, "out_%s.npz" % (variant or "default"))
        env = dict(os.environ, HVN_LIB_VARIANT=variant)
        subprocess.check_call([sys.executable, "-c", CHILD % REPO, f], env=env)
        files[variant] = np.load(f)
    a, b = files[""], files["fullepi"]
    bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))]
    for k in a.files:
        print("%-28s %s %s" % (k, a[k].shape, "EQUAL" if k not in bad else "DIFFERENT (max |d| %g)" % np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()))
    print("operand-set epilogue forms vs the full epilogue: %s" % ("bit-identical on all %d tensors" % len(a.files) if not bad else "MISMATCH in %s" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
