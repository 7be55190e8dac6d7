#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/m16a; mkdir -p $OUT; cd $REPO
for rep in 1 2; do
python scripts/ab_tile.py 512 4096 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt
DMPNN_LIB=$REPO/chemprop_amd/variants/libdmpnn_unrolled.so python scripts/ab_tile.py 512 4096 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt
done
python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu | head -24 | tee $OUT/stamps.txt
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "tile or spill or mega or full_size or module or golden or forward" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-300
