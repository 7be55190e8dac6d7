#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/f16i; mkdir -p $OUT; cd $REPO
python scripts/ab_tile.py 512 4096 2>&1 | grep -v amdgpu | tee $OUT/ab.txt
DMPNN_LIB=$REPO/chemprop_amd/variants/libdmpnn_nospill.so python scripts/ab_tile.py 512 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt
python scripts/bench_configs.py $OUT/configs.json 2>&1 | grep -v amdgpu.ids | grep now | tee $OUT/configs.txt
echo "== DMPNN_K1_SPLIT=1"
DMPNN_K1_SPLIT=1 python scripts/bench_configs.py $OUT/configs_k1split.json synth40 cgr "zinc-512 h300" 2>&1 | grep "now"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log | cut -c1-300
