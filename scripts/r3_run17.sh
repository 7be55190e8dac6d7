#!/bin/bash
# round 3, call 17: training on the tile plan (DMPNN_F_TILE_PLAN) — remaining model tests, host/wall probe, kernel trace of the fused model step
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run17}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "" > $OUT/summary.txt
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
tr = FusedTrainer(m, lr=1e-4, tile_plan=(sys.argv[1] == "tiles"))
y = torch.randn(512, 1, device=dev)
for i in range(222):
    tr.step(b, y)
torch.cuda.synchronize()
PY
for kind in tiles full; do
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$kind -o model -- python /tmp/model_prof.py $kind > $OUT/prof_$kind.log 2>&1; cd $REPO
f=$(find $OUT/prof_$kind -name "*kernel_stats.csv" | head -1); echo "== $kind plan" | tee -a $OUT/summary.txt; [ -n "$f" ] && cp $f $OUT/model_step_kernel_stats_$kind.csv && head -8 $f | cut -c1-170 | tee -a $OUT/summary.txt
done
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -size +20M -delete
echo "== done" | tee -a $OUT/summary.txt
