#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_tile.log; : > $O
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bench_shapes.py -x -q 2>&1 | tail -4 >> $O
timeout 200 python tools/layer_ms.py > gpurun_out/r03_layers_tile256.txt 2>&1; tail -1 gpurun_out/r03_layers_tile256.txt >> $O
grep -E "^d0|u1.conva.wino_gemm|^conv0" gpurun_out/r03_layers_tile256.txt >> $O
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['conv_ms_per_step'], d['roofline']['frac'])" >> $O
cat $O
