#!/bin/bash
# round 6: the tile kernel's tile loop + the estimated grid on the module's steady path
TAG=${1:-r06_loop}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
{
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_model.py tests/test_dropout_gpu.py tests/test_atom_mp.py tests/test_mab.py tests/test_reference_class.py tests/test_pipeline.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -3 | cut -c1-220
DMPNN_TILE_WAVES=4 timeout 300 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -p no:cacheprovider -k "fewer_workgroups or half" 2>&1 | tail -1
for i in 1 2; do
for st in f32 f16; do
  DMPNN_STORE=$st timeout 300 python scripts/ab_tile.py 64 256 512 1024 4096 2>&1 | grep -v amdgpu.ids | sed 's/| K0 .* module forward/| module forward/' | sed "s/^/[$st] /"
done
done
for i in 1 2; do python scripts/prof_model_step.py 512 400 2>&1 | grep -v amdgpu.ids | tail -1; done
} 2>&1 | tee $OUT/summary.txt
