#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/r06_prof_model; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python scripts/prof_model_step.py 512 400 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/summary.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o model -- python $REPO/scripts/prof_model_step.py 512 300 > $OUT/prof.log 2> $OUT/prof.err
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -20 $f | cut -c1-170; done | tee -a $OUT/summary.txt
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete
