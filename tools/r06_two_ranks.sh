cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
HVN_BENCH_SHARED_GPU=1 MASTER_ADDR=127.0.0.1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --fit-steps 12 --no-cpu-baseline --no-roofline --wsi-size 2048 --sustain-seconds 1 > gpurun_out/r06_two_ranks_shared_gpu_default_legs.json 2> gpurun_out/r06_two_ranks.err; echo rc=$?
tail -3 gpurun_out/r06_two_ranks.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_two_ranks_shared_gpu_default_legs.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], list(d.get("variants", {})), d["config"].get("per_rank"))
w = d["variants"]["wsi_2k"]; print({k: w[k] for k in ("patches", "stage1_s", "stage2_s", "instances")}, [(p["patches"], round(p["halo_exchange_s"], 4), p["halo_rows"]) for p in w["per_rank"]])
PY
