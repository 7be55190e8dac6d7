#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/f16g; mkdir -p $OUT; cd $REPO
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "per_step_fused_route or full_size or spill" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log | cut -c1-300
python scripts/bench_configs.py $OUT/configs.json 2>&1 | grep -v amdgpu.ids | tee $OUT/configs.txt
echo "== K1 on the update kernel everywhere (DMPNN_K1_SPLIT=1)"
DMPNN_K1_SPLIT=1 python scripts/bench_configs.py $OUT/configs_k1split.json synth40 cgr-512 "zinc-512 h300" 2>&1 | grep "now"
