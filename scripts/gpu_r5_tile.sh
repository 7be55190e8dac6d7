#!/bin/bash
# round 5: A/B of tile-kernel / K0 variants at the headline shape (scripts/ab_tile.py: tile kernel alone, K0 alone, module forward)
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_r5_tile.sh <tag> variant ...'
TAG=${1:-r05_tile}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=chemprop_amd/variants
{
timeout 200 python scripts/ab_tile.py 256 512 1024 4096 2>&1 | grep -v amdgpu.ids
for v in "$@"; do DMPNN_LIB=$V/libdmpnn_$v.so timeout 200 python scripts/ab_tile.py 256 512 1024 4096 2>&1 | grep -v amdgpu.ids; done
timeout 200 python scripts/ab_tile.py 512 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee $OUT/summary.txt
