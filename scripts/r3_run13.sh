#!/bin/bash
# round 3, call 13: head changes (narrow GEMM for short operands, single-workgroup bounds, batch-norm kernels, device counter, ABI v9),
# FusedTrainer.prefetch_plan: head / model / ffn / agg tests, then the bench legs + a kernel trace of the model step
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run13}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -9 | tee $OUT/summary.txt
# box identity: partition modes, clocks, power cap, and the plain HBM rates (copy / fill of 1 GiB) — the "slow box" question
(rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30) | tee -a $OUT/summary.txt
python - <<PY 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
import torch
dev = torch.device("cuda:0")
a = torch.empty(256 << 20, dtype=torch.float32, device=dev); b = torch.empty_like(a)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
tc = t(lambda: b.copy_(a)); tf = t(lambda: a.fill_(1.0)); tr = t(lambda: a.sum())
print(f"HBM 1 GiB: copy {2 * a.numel() * 4 / tc / 1e12:.2f} TB/s (r+w)   fill {a.numel() * 4 / tf / 1e12:.2f} TB/s (w)   sum {a.numel() * 4 / tr / 1e12:.2f} TB/s (r)")
s = torch.empty(16 << 20, dtype=torch.float32, device=dev)   # 64 MiB: the kept tensors of one training forward
tf2 = t(lambda: s.fill_(1.0), 50)
print(f"fill 64 MiB: {s.numel() * 4 / tf2 / 1e12:.2f} TB/s")
props = torch.cuda.get_device_properties(0)
print("device:", props.name, "CUs", props.multi_processor_count, "mem GiB", round(props.total_memory / 2**30, 1))
PY
timeout 900 python -m pytest tests/test_model.py tests/test_agg.py tests/test_ffn.py tests/test_abi.py tests/test_optim.py tests/test_parity_gpu.py -q -m gpu -k 'not zinc_shape' -p no:cacheprovider -x > $OUT/pytest_head.log 2>&1; echo "pytest(head) rc=$?" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_head.log | tail -15 | cut -c1-300 | tee -a $OUT/summary.txt
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-large-batches > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("fwd us", d["ms_per_step"] * 1e3, "value", d["value"])
print("train_step", d.get("train_step"))
print("model_step", {k: v for k, v in d.get("model_step", {}).items() if k not in ("note", "model", "plan")})
PY
cat > /tmp/model_prof.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from chemprop_amd import agg as cagg, synth
from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
b = synth.random_batch(512, "qm9", seed=1000); b.to(dev)
torch.manual_seed(0)
m = MPNN(BondMessagePassing(d_h=300), cagg.NormAggregation(), RegressionFFN(n_tasks=1, input_dim=300), batch_norm=True).to(dev).train()
tr = FusedTrainer(m, lr=1e-4)
y = torch.randn(512, 1, device=dev)
import copy
b2 = copy.copy(b); b2.edge_index, b2.rev_edge_index, b2.batch = b.edge_index.clone(), b.rev_edge_index.clone(), b.batch.clone()
pair = [b, b2]
for i in range(220):
    tr.prefetch_plan(pair[(i + 1) & 1])
    tr.step(pair[i & 1], y)
torch.cuda.synchronize()
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o model -- python /tmp/model_prof.py > $OUT/prof.log 2>&1; cd $REPO
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/model_step_kernel_stats.csv && head -32 $f | cut -c1-200 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
