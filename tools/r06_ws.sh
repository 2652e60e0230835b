#!/bin/bash
# round 6: floating weight-gradient launches on a second stream per section vs list order (HVN_TRAIN_WGRAD_STREAM=0), one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_wgrad_stream.log; : > $O
timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider -k "first_writer or deterministic or workspace or two_rank or autograd or optimizer" 2>&1 | tail -5 >> $O
J=gpurun_out/r06_wgrad_stream_ab.jsonl; : > $J
for d in 0 1 0 1; do
  HVN_TRAIN_WGRAD_STREAM=$d timeout 400 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | grep "^{" | sed "s/^/HVN_TRAIN_WGRAD_STREAM=$d /" >> $J
done
python - >> $O <<PY
import json
for l in open("$J"):
    i = l.index("{"); tag, d = l[:i], json.loads(l[i:])
    print(tag, "phase", d.get("phase"), "batch", d.get("batch"), "ms/step %.2f" % d.get("ms_per_step", 0), {k: round(v, 2) for k, v in d.items() if k.endswith("_ms")})
PY
cat $O
