#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/fit_synth.py 240 1e-3 2>&1 | grep -v amdgpu.ids | tail -16
