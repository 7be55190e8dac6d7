#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
mkdir -p gpurun_out/r06_wide
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -p no:cacheprovider -k "hpopt_hidden_widths" 2>&1 | grep -v "amdgpu.ids\|^  File" | tail -40 | cut -c1-300 | tee gpurun_out/r06_wide/summary.txt
