#!/bin/bash
# A/B of the two-workgroups-per-CU build of the inference tile kernel + rocprofv3 of the training step on this box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3occ2}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=$REPO/chemprop_amd/variants/libdmpnn_occ2.so
for rep in 1 2; do
timeout 200 python scripts/ab_tile.py 512 1024 4096 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
DMPNN_LIB=$V timeout 200 python scripts/ab_tile.py 512 1024 4096 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
DMPNN_LIB=$V timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "not train and not backward and not grad" > $OUT/pytest_occ2.log 2>&1; echo "pytest(occ2) rc=$?" | tee -a $OUT/ab.txt
tail -4 $OUT/pytest_occ2.log | cut -c1-300 | tee -a $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $REPO/bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches > $OUT/prof_train.json 2> $OUT/prof_train.err
for f in $(find $OUT/prof_train -name "*kernel_stats.csv"); do head -24 $f | cut -c1-220 | tee -a $OUT/ab.txt; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
echo "== done" | tee -a $OUT/ab.txt
