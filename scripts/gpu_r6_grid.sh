#!/bin/bash
# round 6: the tile launch bounded by the molecule count (module's steady path) — loud-guard test, module forward over sizes
TAG=${1:-r06_grid}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
{
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_reference_class.py tests/test_pipeline.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -3 | cut -c1-220
for i in 1 2; do
for st in f32 f16; do
  DMPNN_STORE=$st timeout 300 python scripts/ab_tile.py 64 256 512 1024 2>&1 | grep -v amdgpu.ids | sed 's/| K0 .* module forward/| module forward/' | sed "s/^/[$st] /"
done
done
} 2>&1 | tee $OUT/summary.txt
