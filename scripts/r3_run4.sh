#!/bin/bash
# round 3, fourth GPU call: store-pattern micro-benchmark (slow-box hypothesis), full GPU tests, model step timing + profile
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run4}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 120 ./scripts/micro/store_pattern 2>&1 | tee $OUT/store_pattern.txt | tee $OUT/summary.txt
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_gpu.log | tail -30 | cut -c1-400 | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-batches > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "train", d.get("train_step",{}).get("ms_per_step"), "model_step", {k: v for k, v in (d.get("model_step") or {}).items() if k not in ("note", "model")})
PY
cat > /tmp/model_prof.py <<PY
import sys, time, torch
sys.path.insert(0, "$REPO")
from chemprop_amd import agg as cagg, synth
from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
bmg = synth.random_batch(512, "qm9", seed=1000); bmg.to(dev)
torch.manual_seed(0)
m = MPNN(BondMessagePassing(), cagg.NormAggregation(), RegressionFFN(n_tasks=1), batch_norm=True).to(dev).train()
tr = FusedTrainer(m, lr=1e-4)
y = torch.randn(512, 1, device=dev)
for _ in range(20): tr.step(bmg, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): tr.step(bmg, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"fused model step: enqueue {5e3*(t1-t0):.1f} us/step, total {5e3*(t2-t0):.1f} us/step")
PY
python /tmp/model_prof.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_model -o model -- python /tmp/model_prof.py > $OUT/prof_model.log 2>&1
for f in $(find $OUT/prof_model -name "*kernel_stats.csv"); do head -30 $f | cut -c1-170 | tee -a $OUT/summary.txt; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
echo "== done" | tee -a $OUT/summary.txt
