#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
for rep in 1 2; do
for lib in "" $REPO/chemprop_amd/variants/libdmpnn_prev.so; do
DMPNN_LIB=$lib python bench.py --steps 100 --warmup 10 --mode train --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=%s train step %.1f us  %.1f M/s'%(os.environ.get('DMPNN_LIB','in-tree')[-20:], d['ms_per_step']*1e3, d['value']))"
done; done
