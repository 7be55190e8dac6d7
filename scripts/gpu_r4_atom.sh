#!/bin/bash
# round 4: training of the atom-message blocks (AtomMessagePassing, MAB*) on the tile kernels — their tests, then a timing of the
# atom block's training step beside the bond block's, and rocprofv3 kernel stats of it
TAG=${1:-r4atom}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_atom_mp.py tests/test_mab.py tests/test_abi.py -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest.log | grep "^E  \|passed\|failed\|FAILED\|Error" | head -60 | cut -c1-400 | tee -a $OUT/summary.txt
timeout 300 python scripts/time_atom_blocks.py 2>$OUT/time.err | tee $OUT/atom_time.json | tee -a $OUT/summary.txt
tail -3 $OUT/time.err | cut -c1-300 | tee -a $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o atom -- python $REPO/scripts/time_atom_blocks.py atom > /dev/null 2>&1
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -14 $f | cut -c1-200 | tee -a $OUT/summary.txt; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
