#!/bin/bash
# rocprofv3 kernel stats of the QM9 training step at 4096 molecules (tile kernels on a full plan with molecule tiles)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/trainbig; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $REPO/bench.py --mode train --kind qm9 --mols ${MOLS:-4096} --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-large-batches > /dev/null 2>&1
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -24 $f | cut -c1-170 | tee $OUT/kernel_stats_${MOLS:-4096}.txt; done
