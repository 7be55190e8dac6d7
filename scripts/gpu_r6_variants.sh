#!/bin/bash
# round 6: A/B of build variants of the 8-wave tile kernel on one box (scripts/ab_tile.py), in-tree first and last
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r6_variants.sh <tag> variant ...'
TAG=${1:-r06_variants}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
SIZES=${SIZES:-256 512}
W=${W:-8}
V=chemprop_amd/variants
{
DMPNN_TILE_WAVES=$W timeout 200 python scripts/ab_tile.py $SIZES 2>&1 | grep -v amdgpu.ids
for v in "$@"; do DMPNN_TILE_WAVES=$W DMPNN_LIB=$V/libdmpnn_$v.so timeout 200 python scripts/ab_tile.py $SIZES 2>&1 | grep -v amdgpu.ids; done
DMPNN_TILE_WAVES=$W timeout 200 python scripts/ab_tile.py $SIZES 2>&1 | grep -v amdgpu.ids
echo "--- phase stamps, in-tree"
DMPNN_TILE_WAVES=$W timeout 100 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | head -23
} 2>&1 | tee $OUT/summary.txt
