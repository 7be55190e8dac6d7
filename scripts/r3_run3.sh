#!/bin/bash
# round 3, third GPU call: the whole-model training step (f4) — tests, smoke, bench leg
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run3}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee $OUT/summary.txt
tail -8 $OUT/smoke.log | cut -c1-300 | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_model.py tests/test_abi.py tests/test_distributed.py tests/test_optim.py -q -m gpu -p no:cacheprovider > $OUT/pytest_model.log 2>&1; echo "pytest(model) rc=$?" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_model.log | tail -40 | cut -c1-400 | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-large-batches > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "train", d.get("train_step",{}).get("ms_per_step"), "model_step", d.get("model_step"))
PY
tail -3 $OUT/bench.err | cut -c1-300 | tee -a $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
cat > /tmp/model_prof.py <<PY
import sys, torch
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
for _ in range(60): tr.step(bmg, y)
torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_model -o model -- python /tmp/model_prof.py > $OUT/prof_model.log 2>&1
for f in $(find $OUT/prof_model -name "*kernel_stats.csv"); do head -28 $f | cut -c1-180 | tee -a $OUT/summary.txt; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
echo "== done" | tee -a $OUT/summary.txt
