#!/bin/bash
# round 6: quick check of a k_wgrad16r build — parity of the split-row products, then kernel times of the training step on the rows route
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/${1:-r06_wgradr_quick}; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_model.py -q -m gpu -x -p no:cacheprovider -k "split_row or rows or lean or at_size or train" 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -4 | cut -c1-220 | tee $OUT/parity.txt
cd /tmp
DMPNN_KEEP_ROWS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o train -- python $REPO/bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches > $OUT/prof_train.json 2> $OUT/prof_train.err
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -9 $f | cut -c1-150; done | tee $OUT/kernels.txt
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete
cd $REPO
for k in 1 0; do DMPNN_KEEP_ROWS=$k python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('keep_rows=$k', 'train', d.get('train_step', {}).get('ms_per_step'), 'model', d.get('model_step', {}).get('fused_ms_per_step'))" | tee -a $OUT/kernels.txt; done
