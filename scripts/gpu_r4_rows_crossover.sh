#!/bin/bash
# round 4: where does keeping the tile kernel's messages as split rows start to pay?  training step at 768 .. 3072 molecules, both forms
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/${1:-r4cross}; mkdir -p $OUT
for m in 768 1024 1536 2048 3072; do
for rows in 1 0; do
DMPNN_KEEP_ROWS=$rows timeout 200 python bench.py --steps 60 --warmup 10 --mode train --mols $m --no-cpu-baseline --no-large-batches --no-graph 2>$OUT/err_${m}_$rows.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mols $m KEEP_ROWS=$rows train step %.1f us  (%d message rows)'%(d['ms_per_step']*1e3, 2*d['config']['directed_edges_per_gpu']))" | tee -a $OUT/summary.txt
done
done
