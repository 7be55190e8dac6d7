#!/bin/bash
# round 6: training legs with the tile kernels (forward-keep and backward) as 8-wave workgroups against the 4-wave forms, same box
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r6_train8.sh <tag>'
TAG=${1:-r06_train8}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
{
echo "--- gradient parity, DMPNN_TILE_WAVES=8"
DMPNN_TILE_WAVES=8 timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_atom_mp.py tests/test_dropout_gpu.py tests/test_mab.py tests/test_model.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -4 | cut -c1-220
for w in 4 8 4 8; do
  DMPNN_TILE_WAVES=$w python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('waves $w', 'fwd ms', d.get('ms_per_step'), 'train', d.get('train_step', {}).get('ms_per_step'), 'model', d.get('model_step', {}).get('fused_ms_per_step'))"
done
} 2>&1 | tee $OUT/summary.txt
