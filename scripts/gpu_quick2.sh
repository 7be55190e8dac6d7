#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/quick2
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider -k "grad or train or backward or optim or overfit or pipeline or mab or reference or ffn or mlp or linear or atom or agg" 2>&1 | grep -v "^  File\|^Extension modules" | tail -8 | cut -c1-250 | tee gpurun_out/quick2/pytest.txt
bash scripts/gpu_train40.sh 2>&1 | head -12 | cut -c1-170 | tee -a gpurun_out/quick2/pytest.txt
