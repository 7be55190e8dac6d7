#!/bin/bash
# repeat one test N times in fresh processes (a flaky NaN in the module path's block gradients, round 6)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
K=${1:-"at_size_vs_restatement_and_module_path and 512-qm9-True-norm-1-elu"}
N=${2:-30}
for i in $(seq 1 $N); do
  python -m pytest tests/test_model.py -q -m gpu -p no:cacheprovider -k "$K" 2>&1 | grep -E "passed|failed|AssertionError: " | cut -c1-700 | tr '\n' ' '; echo " [run $i]"
done
