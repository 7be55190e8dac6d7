#!/bin/bash
# round 4, last measurements: host issue time of the whole-model step; kernel stats of the tile kernels' training step at 4 096 molecules
TAG=${1:-r4tail}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 300 python scripts/probe_model_host.py > $OUT/host.txt 2>&1; head -8 $OUT/host.txt | tee $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $REPO/bench.py --mode train --mols 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-graph --no-large-batches > $OUT/bench4096.json 2>$OUT/bench4096.err
python -c "
import json
d=json.loads(open('$OUT/bench4096.json').read().strip().splitlines()[-1]); print('qm9-4096 train step %.1f us'%(d['ms_per_step']*1e3))" | tee -a $OUT/summary.txt
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -12 $f | cut -c1-170 | tee -a $OUT/summary.txt; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
