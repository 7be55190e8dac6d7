#!/bin/bash
# training step of 40-atom molecules (BASELINE configs[3] is a TRAINING workload): bench line + rocprofv3 kernel stats
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/train40; mkdir -p $OUT
for m in 512 4096; do
timeout 600 python bench.py --steps 30 --warmup 5 --mode train --kind synth40 --mols $m --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('synth40-$m train step %.1f us  %.1f M edge-updates/s  route=%s'%(d['ms_per_step']*1e3, d['value'], d.get('route')))" | tee -a $OUT/out.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $REPO/bench.py --mode train --kind synth40 --mols 4096 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-large-batches > /dev/null 2>&1
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -16 $f | cut -c1-150 | tee -a $OUT/out.txt; done
