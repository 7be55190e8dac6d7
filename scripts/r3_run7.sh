#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run7}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -8 | tee $OUT/summary.txt
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_gpu.log | tail -25 | cut -c1-300 | tee -a $OUT/summary.txt
timeout 400 python scripts/ab_configs.py train 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
DMPNN_FUSED16=0 timeout 400 python scripts/ab_configs.py train 2>&1 | grep -v amdgpu.ids | sed 's/^/[DMPNN_FUSED16=0] /' | tee -a $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
cat > /tmp/train40.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from chemprop_amd import distributed as ddp, synth
from chemprop_amd.nn import BondMessagePassing
from chemprop_amd.optim import FlatAdam
dev = torch.device("cuda:0")
b = synth.random_batch(4096, "synth40", seed=1); b.to(dev)
torch.manual_seed(0)
m3 = BondMessagePassing().to(dev).train()
s3 = ddp.GradSync(list(m3.parameters()), modules=[m3]); o3 = FlatAdam(s3, lr=1e-4)
G3 = torch.randn(int(b.V.shape[0]), 300, device=dev)
for _ in range(12):
    with ddp.backward_on_calling_thread():
        o = m3(b); o.backward(G3)
    s3.allreduce(); o3.step()
torch.cuda.synchronize()
print("route", o.grad_fn.st.route if hasattr(o.grad_fn, "st") else None)
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof40 -o t40 -- python /tmp/train40.py > $OUT/prof40.log 2>&1
for f in $(find $OUT/prof40 -name "*kernel_stats.csv"); do head -22 $f | cut -c1-200 | tee -a $OUT/summary.txt; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
echo "== done" | tee -a $OUT/summary.txt
