#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ab1; mkdir -p $OUT; cd $REPO
{
for rep in 1 2; do
python scripts/ab_tile.py 512 4096
DMPNN_LIB=$REPO/chemprop_amd/variants/libdmpnn_nospill.so python scripts/ab_tile.py 512 4096
DMPNN_LIB=$REPO/chemprop_amd/variants/libdmpnn_noinline.so python scripts/ab_tile.py 512 4096
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest.log | cut -c1-300
grep -h "mask flips" $OUT/pytest.log
