#!/bin/bash
# round 5: cycle stamps of the head's three kernels inside one fused training step (512 molecules).
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_r5_head_stamps.sh <tag>'
TAG=${1:-r05_head_stamps}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 300 python scripts/probe_head_rows.py stamps 2>&1 | grep -v amdgpu.ids | tee $OUT/summary.txt
