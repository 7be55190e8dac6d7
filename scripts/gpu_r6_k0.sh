#!/bin/bash
# round 6: K0 over several workgroups (k_prepare_tiles_batch_multi) against the single-workgroup kernel (DMPNN_K0_SINGLE=1), same box
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r6_k0.sh <tag>'
TAG=${1:-r06_k0}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
SIZES=${SIZES:-64 256 512 1024}
{
echo "--- tests that plan on the device"
timeout 900 python -m pytest tests/test_collate.py tests/test_parity_gpu.py tests/test_host.py tests/test_model.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -5 | cut -c1-220
for k in 1 0 1 0; do
  echo "--- DMPNN_K0_SINGLE=$k"
  DMPNN_K0_SINGLE=$k timeout 200 python scripts/ab_tile.py $SIZES 2>&1 | grep -v amdgpu.ids
done
echo "--- K0 stamps (multi)"
timeout 100 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | tail -12
} 2>&1 | tee $OUT/summary.txt
