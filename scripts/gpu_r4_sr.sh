#!/bin/bash
# round 4: the tile kernels' training step with every weight-gradient operand as split rows — a test subset, the training bench line,
# rocprofv3 kernel stats of it
TAG=${1:-r4sr}; KEXPR=${2:-"test_model or tile_kernel_split_f16 or reference_class or spill or sign_bits or tile_plan or backward or dropout"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "$KEXPR" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest.log | grep "^E   .*Error\|^E   .*assert \|passed\|failed\|FAILED" | head -40 | cut -c1-300 | tee -a $OUT/summary.txt
for rows in 1 0; do
DMPNN_KEEP_ROWS=$rows timeout 300 python bench.py --steps 100 --warmup 10 --mode train --no-cpu-baseline --no-large-batches 2>$OUT/bench_$rows.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KEEP_ROWS=$rows train step %.1f us'%(d['ms_per_step']*1e3))" | tee -a $OUT/summary.txt
done
export DMPNN_KEEP_ROWS=1   # (the profile below: the split-row form)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $REPO/bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches > /dev/null 2>&1
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -14 $f | cut -c1-180 | tee -a $OUT/summary.txt; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
