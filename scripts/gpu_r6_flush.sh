#!/bin/bash
# round 6: the kept split rows leaving the forward tile kernel as whole rows (flush_rows): parity of the training routes, the whole-model step, stamps
TAG=${1:-r06_flush}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
{
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_model.py tests/test_dropout_gpu.py tests/test_atom_mp.py tests/test_mab.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -4 | cut -c1-220
for i in 1 2; do python scripts/prof_model_step.py 512 400 2>&1 | grep -v amdgpu.ids | tail -1; done
python scripts/prof_model_step.py 1024 400 2>&1 | grep -v amdgpu.ids | tail -1
python scripts/prof_model_step.py 128 400 2>&1 | grep -v amdgpu.ids | tail -1
DMPNN_TILE_WAVES=8 timeout 100 python scripts/probe_stamps.py 512 tiles keep 2>&1 | grep -v amdgpu.ids | head -24
} 2>&1 | tee $OUT/summary.txt
