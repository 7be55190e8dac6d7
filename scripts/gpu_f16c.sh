#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/f16c; mkdir -p $OUT; cd $REPO
timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "per_step_fused_route or plan_is_bit_exact or full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for st in 0 1 2 3 5; do echo "== DMPNN_STAGGER=$st"; DMPNN_STAGGER=$st python scripts/bench_configs.py $OUT/x.json synth40-4096 cgr-512 2>&1 | grep "now"; done
