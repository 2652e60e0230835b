cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 500 python bench.py --steps 5 --no-cpu-baseline --no-traffic --no-cfg3 --no-wsi-leg --no-roofline 2>gpurun_out/r06_final5.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['variants']['train_step']
print('value %.1f' % d['value'], {k: (round(v['ms_per_step'], 2) if isinstance(v, dict) else v[:60]) for k, v in t.items() if k != 'what'})"
tail -2 gpurun_out/r06_final5.err
