#!/usr/bin/env python
"""The whole-model fused training step (dmpnn_train_step) in a loop, for rocprofv3 --kernel-trace --stats:  python scripts/prof_model_step.py [mols] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chemprop_amd import agg as cagg
from chemprop_amd import synth
from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
from chemprop_amd.nn import BondMessagePassing

dev = torch.device("cuda:0")
n_mols = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.manual_seed(0)
model = MPNN(BondMessagePassing(), cagg.NormAggregation(), RegressionFFN(n_tasks=1), batch_norm=True).to(dev).train()
bmg = synth.random_batch(n_mols, "qm9", seed=1000)
bmg.to(dev)
y = torch.randn(n_mols, 1, device=dev)
tr = FusedTrainer(model, lr=1e-4)
for _ in range(20):
    tr.step(bmg, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(bmg, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"fused model step: enqueue {1e6 * (t1 - t0) / steps:.1f} us/step, total {1e6 * (t2 - t0) / steps:.1f} us/step, route {tr.last_route}")
