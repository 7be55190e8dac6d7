#!/bin/bash
# round 6: quick same-box check of the 8-wave tile kernel: tile-route parity, ab_tile timing (4 | 8 waves), phase stamps
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r6_quick.sh <tag>'
TAG=${1:-r06_quick}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
SIZES=${SIZES:-256 512}
V=chemprop_amd/variants
{
echo "--- parity, DMPNN_TILE_WAVES=8"
DMPNN_TILE_WAVES=8 timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -p no:cacheprovider -k "tile or mega or whole" 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -4 | cut -c1-220
for w in 4 8 8; do
  echo "--- DMPNN_TILE_WAVES=$w"
  DMPNN_TILE_WAVES=$w timeout 200 python scripts/ab_tile.py $SIZES 2>&1 | grep -v amdgpu.ids
done
echo "--- phase stamps, 8 waves"
DMPNN_TILE_WAVES=8 timeout 100 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | head -23
if [ -f $V/libdmpnn_metastamps.so ]; then
echo "--- phase stamps with the meta phase split, 8 waves"
DMPNN_LIB=$V/libdmpnn_metastamps.so DMPNN_TILE_WAVES=8 timeout 100 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | head -8
fi
} 2>&1 | tee $OUT/summary.txt
