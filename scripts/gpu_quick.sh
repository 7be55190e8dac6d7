#!/bin/bash
# quick GPU check: a subset of the GPU tests (-k "$1"), then the training-step and forward bench lines
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/quick; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider -k "$1" 2>&1 | grep -v "^  File\|^Extension modules" | tail -25 | cut -c1-300 | tee $OUT/pytest.txt
for rep in 1 2; do
timeout 300 python bench.py --steps 100 --warmup 10 --mode train --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train step %.1f us  %.1f M/s'%(d['ms_per_step']*1e3, d['value']))" | tee -a $OUT/ab.txt
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forward %.2f us  %.1f M/s'%(d['ms_per_step']*1e3, d['value']))" | tee -a $OUT/ab.txt
