#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/f16a; mkdir -p $OUT; cd $REPO
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "per_step_fused_route or full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 $OUT/pytest.log | cut -c1-400
timeout 900 python scripts/bench_configs.py $OUT/configs.json 2>&1 | grep -v amdgpu.ids | tee $OUT/configs.txt
