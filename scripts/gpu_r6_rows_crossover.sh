#!/bin/bash
# round 6: split-row products (DMPNN_KEEP_ROWS=1) against block products (=0) over batch sizes, same box: block training step | whole-model step
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/${1:-r06_rows_crossover}; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for m in ${MOLS:-64 128 256 512 768 1024}; do for k in 0 1 0 1; do DMPNN_KEEP_ROWS=$k python bench.py --mols $m --steps 200 --warmup 20 --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('mols $m keep_rows=$k', 'train', d.get('train_step', {}).get('ms_per_step'), 'model', d.get('model_step', {}).get('fused_ms_per_step'))"; done; done | tee $OUT/summary.txt
