#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run5}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -8 | tee $OUT/summary.txt
V=$REPO/chemprop_amd/variants/libdmpnn_ring16.so
for rep in 1 2; do
timeout 300 python scripts/ab_configs.py train 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
DMPNN_LIB=$V timeout 300 python scripts/ab_configs.py train 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
done
DMPNN_LIB=$V timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -p no:cacheprovider -x > $OUT/pytest_ring.log 2>&1; echo "pytest(ring16) rc=$?" | tee -a $OUT/summary.txt
tail -4 $OUT/pytest_ring.log | cut -c1-300 | tee -a $OUT/summary.txt
timeout 600 python -m pytest tests/test_model.py -q -m gpu -p no:cacheprovider > $OUT/pytest_model.log 2>&1; echo "pytest(model) rc=$?" | tee -a $OUT/summary.txt
tail -6 $OUT/pytest_model.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
