#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_full.log; : > $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 >> $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $O
SECONDS=0
timeout 900 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
echo "bench rc=$? wall=${SECONDS}s" >> $O
python - <<'PY' >> $O 2>&1
import json
d=json.load(open('gpurun_out/r03_bench_default.json'))
print('value', d['value'], 'h2h', d.get('value_host_to_host'), 'ms', d['ms_per_step'])
print('stage', d['config']['stage_ms'])
r=d['roofline']; print('roof', r['frac'], r['conv_ms_per_step'], r['traffic'], r.get('traffic_hbm_bytes_per_step'), r['traffic_unit'][:100])
for k,v in d['variants'].items(): print(k, v['value'], v.get('ms_per_step'))
c=d['variants'].get('cfg3_fast_b64_bf16'); 
if c: print('cfg3 roof', c['roofline']['frac'], c['roofline']['conv_ms_per_step'], c['instances_last_step'])
print('cpu', d.get('cpu_baseline'))
PY
cat $O
