#!/bin/bash
# ONE entry point for GPU-box sessions:  gpurun -- 'bash tools/gpu_run.sh <target> [tag]'.  Outputs land in gpurun_out/ (scratch);
# what is to be judged is copied to profiles/ afterwards.  tools/refresh_profiles.sh is the round-end target set.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-bench}; R=${2:-r05}; O=gpurun_out/${R}_${T}.log; : > $O
bench() {   # bench <name> [env / args ...]: one bench.py run -> gpurun_out/<R>_<name>.json + its summary in the log
  local name=$1; shift
  SECONDS=0
  env "${ENVV[@]}" timeout 900 python bench.py "$@" > gpurun_out/${R}_${name}.json 2> gpurun_out/${R}_${name}.err
  echo "== bench $name (${ENVV[*]} $*) rc=$? wall=${SECONDS}s" >> $O
  python tools/bench_summary.py gpurun_out/${R}_${name}.json >> $O 2>&1 || tail -5 gpurun_out/${R}_${name}.err >> $O
}
ENVV=()
case $T in
  first)      # round-4 first contact: trained-like parity (both inits if the first fails), the default bench line, F(6,5) A/B
    timeout 900 python -m pytest tests/test_gpu_trained_like.py -x -q -s 2>&1 | grep -v "^$" | tail -25 >> $O
    if ! grep -q "2 passed" $O; then
      echo "== retry with HVN_FIT_INIT=synth" >> $O
      HVN_FIT_INIT=synth timeout 900 python -m pytest tests/test_gpu_trained_like.py -x -q -s 2>&1 | grep -v "^$" | tail -25 >> $O
    fi
    bench bench_default
    ENVV=(HVN_WINOGRAD=6); bench bench_f65 --steps 10 --no-cpu-baseline --no-variants --no-traffic --checkpoint random
    ENVV=(); bench bench_random --steps 10 --no-cpu-baseline --no-variants --no-traffic --checkpoint random
    ;;
  bench)
    bench bench_default
    ;;
  sched)      # launch-schedule A/B on one box: sub-batch streams x decoder lanes x where the tile selection is timed
    Q="--steps 20 --no-cpu-baseline --no-variants --no-roofline --checkpoint random"
    for cfg in "HVN_SPLIT=1 HVN_LANES=0" "HVN_SPLIT=2 HVN_LANES=2" "HVN_SPLIT=2 HVN_LANES=2 HVN_TUNE_SUB=0" "HVN_SPLIT=4 HVN_LANES=2" \
               "HVN_SPLIT=4 HVN_LANES=2 HVN_TUNE_SUB=0" "HVN_SPLIT=2 HVN_LANES=2 HVN_SPLIT_DECODER=1" "HVN_SPLIT=3 HVN_LANES=2" "HVN_SPLIT=2 HVN_LANES=0"; do
      ENVV=($cfg); bench sched_$(echo $cfg | tr -d ' =A-Z_') $Q
    done
    grep -E "^== bench|^value" $O > gpurun_out/${R}_sched_ab.txt
    ;;
  x3)         # the bf16x3 kernel: unit tests, goldens, trained-like margins, then speed (native | 9 terms | 6 terms) on one box
    timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -s 2>&1 | grep -v "^$" | tail -12 >> $O
    timeout 900 python -m pytest tests/test_gpu_trained_like.py -x -q -s 2>&1 | grep -v "^$" | tail -12 >> $O
    Q="--steps 10 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
    for cfg in "HVN_X3=0" "HVN_X3=9" "HVN_X3=6"; do ENVV=($cfg); bench x3_$(echo $cfg | tr -d ' =A-Z_') $Q; done
    ;;
  layers)     # per-launch tables, native vs bf16x3 (6 terms), one box
    for x in 0 6 9; do HVN_X3=$x timeout 300 python tools/layer_ms.py > gpurun_out/${R}_layers_x3_$x.txt 2>&1; tail -1 gpurun_out/${R}_layers_x3_$x.txt >> $O; done
    ;;
  core)       # the GPU tests every kernel / lowering change touches + the default bench line + the d1-on-bf16x3 option
    timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_chain.py tests/test_gpu_bench_shapes.py tests/test_gpu_x3.py tests/test_gpu_conv.py -x -q 2>&1 | tail -6 >> $O
    bench bench_default
    Q="--steps 10 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
    for cfg in "HVN_X3_D1=0" "HVN_X3_D1=1"; do ENVV=($cfg); bench d1_$(echo $cfg | tr -d ' =A-Z_') $Q; done
    ;;
  bf16)       # the bf16 conv loop A/B (HVN_BF16_LOOP=0: round 2's loop) + launch schedule, cfg 3, one box; then its parity tests
    Q="--dtype bf16 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
    for cfg in "HVN_BF16_LOOP=0" "HVN_BF16_LOOP=1" "HVN_BF16_LOOP=1 HVN_SPLIT=2 HVN_LANES=2"; do ENVV=($cfg); bench bf16_$(echo $cfg | tr -d ' =A-Z_') $Q; done
    grep -E "^== bench|^value|^roofline" $O > gpurun_out/${R}_bf16_loop_ab.txt
    for x in 0 1; do HVN_BF16_LOOP=$x timeout 200 python tools/layer_ms.py --dtype bf16 --mode fast --nr-types 6 --batch 64 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_layers_cfg3_bf16_loop$x.txt; done
    timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bench_shapes.py -x -q 2>&1 | tail -5 >> $O
    ;;
  train)      # training step: fp32 pipe vs bf16x3 forward / data-gradient products on one box, then the training tests
    for x in 0 6; do HVN_TRAIN_X3=$x timeout 400 python tools/train_bench.py --steps 10 --warmup 3 2>/dev/null | grep "^{" | sed "s/^/HVN_TRAIN_X3=$x /" >> gpurun_out/${R}_train_bench.jsonl; done
    python - >> $O <<'PY'
import json
for l in open("gpurun_out/r04_train_bench.jsonl"):
    tag, js = l.split(" ", 1); d = json.loads(js)
    print(tag, "phase", d.get("phase"), "batch", d.get("batch"), "ms/step %.2f" % d.get("ms_per_step", 0), {k: round(v, 2) for k, v in d.items() if k.endswith("_ms")})
PY
    timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -s 2>&1 | grep -E "passed|failed|Error|assert|bf16x3|direct vs" | tail -14 >> $O
    ;;
  dense)      # dense-unit grouped conv: form 2 (whole patch per tile) vs form 3 (rolling patch, runs of tiles), one box; parity tests first
    timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_net.py -x -q 2>&1 | tail -3 >> $O
    Q="--steps 10 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
    for cfg in "HVN_DENSE_FORM=2" "HVN_DENSE_RUN=4" "HVN_DENSE_RUN=3" "HVN_DENSE_RUN=5" "HVN_DENSE_RUN=8"; do ENVV=($cfg); bench dense_$(echo $cfg | tr -d ' =A-Z_') $Q; done
    grep -E "^== bench|^value|^roofline" $O > gpurun_out/${R}_dense_ab.txt
    for cfg in "HVN_DENSE_FORM=2" "HVN_DENSE_RUN=4"; do env $cfg timeout 200 python tools/layer_ms.py 2>/dev/null | grep "dense.units.*conv2" | head -12 > gpurun_out/${R}_dense_layers_$(echo $cfg | tr -d ' =A-Z_').txt; done
    ;;
  trainprof)  # rocprofv3 kernel summary of the training step, per phase
    for ph in 0 1; do
      CMD="python tools/train_bench.py --steps 4 --warmup 2 --phase $ph"
      timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_tprof$ph -o r -- $CMD > gpurun_out/${R}_train_profiled_phase$ph.json 2>/dev/null
      python tools/kernel_stats.py $(find gpurun_out/${R}_tprof$ph -name "*_results.db" | head -1) "rocprofv3 --kernel-trace --stats -- $CMD" > gpurun_out/${R}_train_kernel_stats_phase$ph.csv 2>/dev/null
      rm -rf gpurun_out/${R}_tprof$ph
      head -16 gpurun_out/${R}_train_kernel_stats_phase$ph.csv | cut -c1-140 >> $O
    done
    ;;
  x3g)        # round 5: the LDS-DMA forms of the bf16x3 kernel (csrc/hvn_conv_x3g.hip): bit-equality tests, per-launch tables with every
              # eligible launch forced onto hvn_conv_x3.hip | the 256-pixel form | the 128-pixel form, then bench lines without / with them
    timeout 900 python -m pytest tests/test_gpu_x3.py -q -k "lds_dma" --tb=line 2>&1 | tail -40 >> $O
    for cfg in "HVN_X3G=0" "HVN_X3G_FORCE=896" "HVN_X3G_FORCE=640"; do
      f=gpurun_out/${R}_layers_$(echo $cfg | tr -d ' =A-Z_').txt
      env $cfg timeout 300 python tools/layer_ms.py > $f 2>&1; echo "== $cfg: $(tail -1 $f)" >> $O
    done
    Q="--steps 10 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
    for cfg in "HVN_X3G=0" "HVN_X3G=1"; do ENVV=($cfg); bench x3g_$(echo $cfg | tr -d ' =A-Z_') $Q; done
    ;;
  chainx3)    # round 5: d0's seams chained on the bf16 pipe (csrc/hvn_conv_chain_x3.hip): parity, goldens, trained-like margins, per-launch tables
              # and bench lines for HVN_X3_CHAIN = "" (fp32-pipe chains, rounds 3-4) | d0 | d0d1
    timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_net.py -q --tb=line 2>&1 | tail -15 >> $O
    Q="--steps 10 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
    for c in "" d0 d0d1; do
      f=gpurun_out/${R}_layers_chain_${c:-fp32}.txt
      HVN_X3_CHAIN=$c timeout 300 python tools/layer_ms.py > $f 2>&1; echo "== HVN_X3_CHAIN=$c: $(tail -1 $f)" >> $O
      grep -E "^d0|^d1.units.0|\+" $f | head -24 >> $O
      ENVV=(HVN_X3_CHAIN=$c); bench chain_${c:-fp32} $Q
    done
    timeout 900 python -m pytest tests/test_gpu_trained_like.py -q -s 2>&1 | grep -v "^$" | tail -25 >> $O
    ;;
  chainx3r)   # round 5: the bf16x3 chain with a register-resident input tile (csrc/hvn_conv_chain_x3r.hip): bit-equality tests (under a
              # short timeout: counted waits + raw barriers), then d0's per-launch rows and bench lines with the form off | forced | timed
    timeout 600 python -m pytest tests/test_gpu_chain.py -q --tb=short -x 2>&1 | tail -15 >> $O
    Q="--steps 10 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
    for c in 0 force 1; do
      f=gpurun_out/${R}_layers_x3r_$c.txt
      HVN_CHAIN_X3R=$c timeout 300 python tools/layer_ms.py > $f 2>&1; echo "== HVN_CHAIN_X3R=$c: $(tail -1 $f)" >> $O
      grep -E "\+" $f | head -6 >> $O
    done
    for c in 0 1; do ENVV=(HVN_CHAIN_X3R=$c); bench x3r_$c $Q; done
    ;;
  winoxcd)    # round 5: XCD-contiguous tile ranges in the Winograd input transform (csrc/hvn_net_ops.hip); A/B against the "noxcd" build
    timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_conv.py -q --tb=short -x -k "not slow" 2>&1 | tail -5 >> $O
    for v in noxcd ""; do
      f=gpurun_out/${R}_layers_winoxcd_${v:-default}.txt
      HVN_LIB_VARIANT=$v timeout 300 python tools/layer_ms.py > $f 2>&1; echo "== HVN_LIB_VARIANT=$v: $(tail -1 $f)" >> $O
      grep "wino_in" $f | awk '{s+=$(NF)} END {print "   wino_in total us:", s}' >> $O
    done
    Q="--steps 10 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
    for v in noxcd ""; do ENVV=(HVN_LIB_VARIANT=$v); bench winoxcd_${v:-default} $Q; done
    ;;
  wgradx3)    # round 5: weight gradients on the bf16 pipe (csrc/hvn_wgrad_x3.hip) + the LDS-DMA conv forms in the training step: kernel tests,
              # then the training step with / without them on one box
    timeout 900 python -m pytest tests/test_gpu_train.py -q --tb=line -k "wgrad" 2>&1 | tail -8 >> $O
    for cfg in "HVN_TRAIN_WGRAD_X3=0 HVN_X3G=0" "HVN_TRAIN_WGRAD_X3=1 HVN_X3G=0" "HVN_TRAIN_WGRAD_X3=1 HVN_X3G=1"; do
      env $cfg timeout 400 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | grep "^{" | sed "s/^/$cfg /" >> gpurun_out/${R}_train_ab.jsonl
    done
    python - >> $O <<PY
import json
for l in open("gpurun_out/${R}_train_ab.jsonl"):
    i = l.index("{"); tag, d = l[:i], json.loads(l[i:])
    print(tag, "phase", d.get("phase"), "batch", d.get("batch"), "ms/step %.2f" % d.get("ms_per_step", 0), {k: round(v, 2) for k, v in d.items() if k.endswith("_ms")})
PY
    ;;
  bf16g)      # round 5: the LDS-DMA form of the bf16 convolution (csrc/hvn_conv_bf16g.hip), cfg 3: bit-equality tests, per-launch tables
              # forced old | 256-pixel | 128-pixel form, bench lines without / with the forms
    timeout 900 python -m pytest tests/test_gpu_bf16.py -q -k "lds_dma" --tb=line 2>&1 | tail -12 >> $O
    A="--dtype bf16 --mode fast --nr-types 6 --batch 64"
    for cfg in "HVN_BF16G=0" "HVN_BF16G_FORCE=896" "HVN_BF16G_FORCE=640"; do
      f=gpurun_out/${R}_layers_cfg3_$(echo $cfg | tr -d ' =A-Z_').txt
      env $cfg timeout 300 python tools/layer_ms.py $A 2>/dev/null | grep -v amdgpu.ids > $f; echo "== $cfg: $(tail -1 $f)" >> $O
    done
    Q="$A --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
    for cfg in "HVN_BF16G=0" "HVN_BF16G=1"; do ENVV=($cfg); bench bf16g_$(echo $cfg | tr -d ' =A-Z_') $Q; done
    ;;
  trained)
    timeout 900 python -m pytest tests/test_gpu_trained_like.py -x -q -s 2>&1 | grep -v "^$" | tail -25 >> $O
    ;;
  tests)
    timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 >> $O
    ;;
  *) echo "unknown target $T" >> $O ;;
esac
cat $O
