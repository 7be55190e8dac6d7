#!/bin/bash
# round 6: the training step on split-row products (DMPNN_KEEP_ROWS=1, k_wgrad16r) against the block products (k_wsplit16 + k_wgrad16), same box;
# DMPNN_WGRADR_WGS = workgroups of the product launch (default: one per CU — wgrad16r_rows_per_split)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r06_wgradr}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); oc = d.get('other_configs', {})
print('$1', 'train', d.get('train_step', {}).get('ms_per_step'), 'model', d.get('model_step', {}).get('fused_ms_per_step'), {k: v.get('train_step_us') for k, v in oc.items() if 'train_step_us' in v})"; }
{
echo "--- parity of everything that multiplies split rows"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_model.py tests/test_atom_mp.py tests/test_mab.py -q -m gpu -x -p no:cacheprovider -k "split_row or rows or lean or fused16 or at_size or train" 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -4 | cut -c1-220
DMPNN_KEEP_ROWS=0 run blocks --no-large-batches
DMPNN_KEEP_ROWS=1 run rows --no-large-batches
DMPNN_KEEP_ROWS=1 DMPNN_WGRADR_WGS=128 run rows_wgs128 --no-large-batches
DMPNN_KEEP_ROWS=1 DMPNN_WGRADR_WGS=192 run rows_wgs192 --no-large-batches
DMPNN_KEEP_ROWS=1 DMPNN_WGRADR_WGS=512 run rows_wgs512 --no-large-batches
DMPNN_KEEP_ROWS=0 run blocks --no-large-batches
DMPNN_KEEP_ROWS=1 run rows --no-large-batches
run default_with_large
} 2>&1 | tee $OUT/summary.txt
