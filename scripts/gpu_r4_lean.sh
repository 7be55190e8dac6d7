#!/bin/bash
# round 4: the lean training route of molecules beyond the tile — its tests, then the training step of 40-atom molecules (bench line +
# rocprofv3 kernel stats)
TAG=${1:-r4lean}; KEXPR=${2:-"lean or relu_gradients_at_size or per_step_fused_route"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "$KEXPR" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest.log | grep "^E  \|passed\|failed\|^lean-\|FAILED\|Error" | head -40 | cut -c1-400 | tee -a $OUT/summary.txt
for m in 512 4096; do
timeout 600 python bench.py --steps 30 --warmup 5 --mode train --kind synth40 --mols $m --no-cpu-baseline --no-large-batches 2>$OUT/bench_$m.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('synth40-$m train step %.1f us  %.1f M edge-updates/s  route=%s'%(d['ms_per_step']*1e3, d['value'], d.get('route')))" | tee -a $OUT/summary.txt
tail -2 $OUT/bench_$m.err | cut -c1-300 | tee -a $OUT/summary.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $REPO/bench.py --mode train --kind synth40 --mols 4096 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-large-batches > /dev/null 2>&1
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -22 $f | cut -c1-170 | tee -a $OUT/summary.txt; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
