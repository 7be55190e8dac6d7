#!/bin/bash
# round 5: timing diagnostics of k_step16 (variants with WRONG numbers: which part of the tile's work bounds the launch?)
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_r5_diag.sh <tag> variant ...'
TAG=${1:-r05_diag}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=chemprop_amd/variants
{
timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do DMPNN_LIB=$V/libdmpnn_$v.so timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids; done
timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee $OUT/summary.txt
