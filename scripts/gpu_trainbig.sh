#!/bin/bash
# training on the tile kernels beyond the single-workgroup plan (dmpnn_prepare_with_batch): the new tests, then the
# step time of QM9-like batches with the full plan carrying molecule tiles (default) and without (DMPNN_TRAIN_TILES=0)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/trainbig; mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "full_plan_with_molecule_tiles or relu_gradients_at_size or backward_full_size or validate_modes or tile_plan" 2>&1 | tail -15 | tee $OUT/pytest.txt
for tt in 1 0; do for m in 1024 4096; do
DMPNN_TRAIN_TILES=$tt timeout 600 python bench.py --steps 30 --warmup 5 --mode train --kind qm9 --mols $m --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qm9-$m TRAIN_TILES=$tt train step %.1f us  %.1f M edge-updates/s  route=%s'%(d['ms_per_step']*1e3, d['value'], d.get('route')))" | tee -a $OUT/out.txt
done; done
