#!/bin/bash
# smoke() + the whole GPU test suite, nothing else -> gpurun_out/tests_only/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
mkdir -p gpurun_out/tests_only
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules" | tail -6 | cut -c1-200 | tee gpurun_out/tests_only/pytest.txt
