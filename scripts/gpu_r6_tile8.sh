#!/bin/bash
# round 6: the tile kernel as ONE 512-thread workgroup per tile (DMPNN_TILE_WAVES=8) against the 4-wave form, same box:
# parity of the tile route under the forced 8-wave form, then scripts/ab_tile.py (tile kernel alone, K0 alone, module forward)
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r6_tile8.sh <tag> [variant ...]'   (variants: chemprop_amd/variants/libdmpnn_<v>.so)
TAG=${1:-r06_tile8}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
SIZES=${SIZES:-64 256 512 1024}
V=chemprop_amd/variants
{
echo "--- parity, DMPNN_TILE_WAVES=8"
DMPNN_TILE_WAVES=8 timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_atom_mp.py tests/test_dropout_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -8 | cut -c1-220
for w in 4 8; do
  echo "--- DMPNN_TILE_WAVES=$w"
  DMPNN_TILE_WAVES=$w timeout 200 python scripts/ab_tile.py $SIZES 2>&1 | grep -v amdgpu.ids
done
for v in "$@"; do
  echo "--- DMPNN_TILE_WAVES=8 variant $v"
  DMPNN_TILE_WAVES=8 DMPNN_LIB=$V/libdmpnn_$v.so timeout 200 python scripts/ab_tile.py $SIZES 2>&1 | grep -v amdgpu.ids
done
echo "--- DMPNN_TILE_WAVES=8 (again)"
DMPNN_TILE_WAVES=8 timeout 200 python scripts/ab_tile.py $SIZES 2>&1 | grep -v amdgpu.ids
echo "--- phase stamps, 8 waves"
DMPNN_TILE_WAVES=8 timeout 100 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | head -24
echo "--- phase stamps, 4 waves"
DMPNN_TILE_WAVES=4 timeout 100 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | head -24
} 2>&1 | tee $OUT/summary.txt
