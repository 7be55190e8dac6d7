#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run9}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -20 | tee $OUT/summary.txt
timeout 300 python bench.py --steps 50 --warmup 10 --mode train --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train step %.1f us  %.1f M/s'%(d['ms_per_step']*1e3, d['value']))" | tee -a $OUT/summary.txt
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -m gpu -p no:cacheprovider -k "per_step or fused_route or wide" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_gpu.log | tail -8 | cut -c1-300 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
