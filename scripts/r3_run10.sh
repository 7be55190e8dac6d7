#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run10}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -9 | tee $OUT/summary.txt
timeout 300 python -m pytest tests/test_dropout_gpu.py -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_gpu.log | tail -25 | cut -c1-300 | tee -a $OUT/summary.txt
cat > /tmp/drop_time.py <<PY
import sys, time, torch
sys.path.insert(0, "$REPO")
from chemprop_amd import distributed as ddp, synth
from chemprop_amd.nn import BondMessagePassing
from chemprop_amd.optim import FlatAdam
dev = torch.device("cuda:0")
b = synth.random_batch(512, "qm9", seed=1000); b.to(dev)
G = torch.randn(int(b.V.shape[0]), 300, device=dev)
for p, act in ((0.0, "relu"), (0.2, "relu"), (0.2, "tanh")):
    torch.manual_seed(0)
    m = BondMessagePassing(dropout=p, activation=act).to(dev).train()
    s = ddp.GradSync(list(m.parameters()), modules=[m]); o = FlatAdam(s, lr=1e-4)
    def step():
        with ddp.backward_on_calling_thread():
            out = m(b); r = out.grad_fn.st.route if hasattr(out.grad_fn, "st") else "rows"; out.backward(G)
        s.allreduce(); o.step()
        return r
    for _ in range(10): out = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    route = out
    print(f"train step p={p} act={act}: {(t1 - t0) * 1e4:.1f} us  route={route}")
PY
python /tmp/drop_time.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
