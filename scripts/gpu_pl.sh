#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/pl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o large -- python $REPO/scripts/prof_large.py --kind synth40 --mols 4096 > $OUT/prof_large.json 2> $OUT/prof.err
cat $OUT/prof_large.json
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -12 $f | cut -c1-160; done
