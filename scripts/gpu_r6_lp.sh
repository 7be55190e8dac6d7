#!/bin/bash
# round 6: the tile kernel on the hi halves alone (DMPNN_STORE=f16 on the mega16 route): parity at its own bar, timing beside the exact form
TAG=${1:-r06_lp}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -p no:cacheprovider -s -k "half" 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | grep "half operands\|passed\|failed\|Error\|assert" | cut -c1-220
for st in f32 f16 f32 f16; do
  echo "--- DMPNN_STORE=$st"
  DMPNN_STORE=$st timeout 200 python scripts/ab_tile.py 256 512 1024 2>&1 | grep -v amdgpu.ids
done
echo "--- phase stamps, DMPNN_STORE=f16"
DMPNN_STORE=f16 DMPNN_TILE_WAVES=8 timeout 100 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | head -23
} 2>&1 | tee $OUT/summary.txt
