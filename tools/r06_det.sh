#!/bin/bash
# round 6, first GPU session: deterministic training reduce -- tests, step-time A/B against the atomic form, then the bf16-vs-fp32 PQ table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_det.log; : > $O
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider -k "deterministic or workspace or two_fits or losses or head or conv0 or wgrad" 2>&1 | tail -15 >> $O
for d in 0 1; do
  HVN_TRAIN_DETERMINISTIC=$d timeout 400 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | grep "^{" | sed "s/^/HVN_TRAIN_DETERMINISTIC=$d /" >> gpurun_out/r06_train_det_ab.jsonl
done
python - >> $O <<PY
import json
for l in open("gpurun_out/r06_train_det_ab.jsonl"):
    i = l.index("{"); tag, d = l[:i], json.loads(l[i:])
    print(tag, "phase", d.get("phase"), "batch", d.get("batch"), "ms/step %.2f" % d.get("ms_per_step", 0), {k: round(v, 2) for k, v in d.items() if k.endswith("_ms")})
PY
timeout 900 python tools/bf16_pq_table.py --seeds 0,1,2,3,4,5,6,7 --tiles 48 > gpurun_out/r06_bf16_pq_table.txt 2>&1
tail -40 gpurun_out/r06_bf16_pq_table.txt >> $O
cat $O
