#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/r06_prof_rows; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp; cd /tmp
DMPNN_KEEP_ROWS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o train -- python $REPO/bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches > $OUT/prof_train.json 2> $OUT/prof_train.err
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -14 $f | cut -c1-150; done
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete
