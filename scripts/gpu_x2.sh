#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/x2; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider -k "half_storage or per_step_fused or full_size or wide or properties or reference or spill or export or mab or atom" 2>&1 | grep -v "^  File\|^Extension modules" | tail -6 | cut -c1-300 | tee $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o large -- python $REPO/scripts/prof_large.py --kind synth40 --mols 4096 > $OUT/prof_large.json 2> $OUT/prof.err
cat $OUT/prof_large.json | tee -a $OUT/pytest.txt
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -7 $f | cut -c1-160 | tee -a $OUT/pytest.txt; done
cd $REPO
python scripts/bench_configs.py 2>/dev/null | tail -12 | tee -a $OUT/pytest.txt
