#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/tr1; mkdir -p $OUT; cd $REPO
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "backward or grad or training or train or smoke or mab or pipeline" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
python bench.py --steps 100 --warmup 10 --mode train --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train step', d['ms_per_step'], 'ms', d['value'], 'M/s', d['config']['launch'])"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $REPO/bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches > $OUT/run.txt 2>&1
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -12 $f | cut -c1-180; done
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete; true
