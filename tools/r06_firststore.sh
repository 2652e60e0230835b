#!/bin/bash
# round 6: first-writer stores in the backward pass (no zero-fill of 73 % of the gradient arena, BN backward without reading a / the old grad z,
# data gradients without the residual operand) vs rounds 3-5 (HVN_TRAIN_FIRST_STORE=0: everything accumulates, everything cleared), one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_first_store.log; : > $O
timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider 2>&1 | tail -5 >> $O
J=gpurun_out/r06_first_store_ab.jsonl; : > $J
for d in 0 1 0 1; do
  HVN_TRAIN_FIRST_STORE=$d timeout 400 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | grep "^{" | sed "s/^/HVN_TRAIN_FIRST_STORE=$d /" >> $J
done
python - >> $O <<PY
import json
for l in open("$J"):
    i = l.index("{"); tag, d = l[:i], json.loads(l[i:])
    print(tag, "phase", d.get("phase"), "batch", d.get("batch"), "ms/step %.2f" % d.get("ms_per_step", 0), {k: round(v, 2) for k, v in d.items() if k.endswith("_ms")})
PY
cat $O
