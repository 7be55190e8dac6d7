#!/bin/bash
TAG=${1:-dbg}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for k in "512-qm9" "4096-qm9" "512-synth40" "512-zinc"; do
  echo "=== full_size $k" >> $OUT/dbg.log
  timeout 300 python -m pytest tests/test_parity_gpu.py -q -x -s -p no:cacheprovider -k "full_size_vs_oracle and $k and not backward" >> $OUT/dbg.log 2>&1
  echo "rc=$?" >> $OUT/dbg.log
done
echo "=== rest" >> $OUT/dbg.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -p no:cacheprovider -k "backward or frozen or overfit or properties or custom or cpu_tensors or invalid_vd" >> $OUT/dbg.log 2>&1
echo "rc=$?" >> $OUT/dbg.log
grep -v "^  File\|^$" $OUT/dbg.log | tail -80
