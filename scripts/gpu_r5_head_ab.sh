#!/bin/bash
# round 5: the A/B of scripts/probe_head_rows.py alone (fused step / module path under the head's forms), 512 and 64 molecules.
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_r5_head_ab.sh <tag>'
TAG=${1:-r05_head_ab}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
{
timeout 300 python scripts/probe_head_rows.py 2>&1 | grep -v "amdgpu.ids\|step-1 gradient"
MOLS=64 timeout 300 python scripts/probe_head_rows.py 2>&1 | grep -v "amdgpu.ids\|step-1 gradient" | tail -12
} 2>&1 | tee $OUT/summary.txt
