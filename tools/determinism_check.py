"""Run-to-run and form-to-form bit equality of the whole network under load (python tools/determinism_check.py [batch] [runs]).

Every kernel form the engine may pick for a launch is meant to produce the same bits (tests/test_gpu_x3.py, test_gpu_chain.py compare
them per launch shape and once per network at batch 2).  This tool repeats the comparison at a batch that fills the chip, several runs
per configuration: a synchronisation slip in a kernel (a counted s_waitcnt that is one short, a missing barrier) shows up as a
run-to-run or form-to-form difference only when the memory system is busy.  Prints one line per configuration; exit code 1 on any difference."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd import net_desc  # noqa: E402
from hover_net_amd.synth import synth_state_dict, synth_tiles  # noqa: E402

CONFIGS = [
    ("staged forms only", {"HVN_X3G": "0", "HVN_CHAIN_X3R": "0"}),
    ("x3g 128-row form forced", {"HVN_X3G_FORCE": "640", "HVN_CHAIN_X3R": "0"}),
    ("x3g 256-row form forced", {"HVN_X3G_FORCE": "896", "HVN_CHAIN_X3R": "0"}),
    ("chain x3r forced", {"HVN_X3G": "0", "HVN_CHAIN_X3R": "force"}),
    ("default (timed choices)", {}),
]
KEYS = ("HVN_X3G", "HVN_X3G_FORCE", "HVN_CHAIN_X3R")


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tiles = torch.from_numpy(synth_tiles(batch, 270, seed=3)).cuda()
    sd = synth_state_dict("original", 5, seed=2)
    ref, bad = None, 0
    for name, env in CONFIGS:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        net = net_desc.create_model(mode="original", nr_types=5, input_ch=3)
        net.load_state_dict(sd, strict=True)
        net.max_batch = batch
        net = net.cuda().eval()
        eng = net.engine(batch)
        outs = []
        for _ in range(runs):
            logits, _pred = eng.run(tiles)
            torch.cuda.synchronize()
            outs.append({k: v.clone() for k, v in logits.items()})
        same_runs = all(torch.equal(outs[0][k], o[k]) for o in outs[1:] for k in o)
        if ref is None:
            ref = outs[0]
        same_ref = all(torch.equal(ref[k], outs[0][k]) for k in ref)
        worst = max(float((ref[k] - o[k]).abs().max()) for o in outs for k in o)
        print("%-28s run-to-run %s   vs staged forms %s   (max |diff| %.3g)" % (name, "equal" if same_runs else "DIFFERENT", "equal" if same_ref else "DIFFERENT", worst), flush=True)
        bad += (not same_runs) + (not same_ref)
        del net, eng
        torch.cuda.empty_cache()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
