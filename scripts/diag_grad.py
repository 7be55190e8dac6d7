"""Diagnostic: parameter gradients of one training step against fp64 / fp32 autograd of the reference's op sequence.
usage: python scripts/diag_grad.py <kind> <n_mols> <act> [seed]   (routes are chosen by the DMPNN_* environment)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chemprop_amd import synth  # noqa: E402
from chemprop_amd.nn import BondMessagePassing  # noqa: E402
from oracle import dmpnn_torch as ot  # noqa: E402

kind, n_mols, act = sys.argv[1], int(sys.argv[2]), sys.argv[3]
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 13
dims = dict(d_v=106, d_e=28) if kind == "cgr" else {}
bmg = synth.random_batch(n_mols, kind, seed=seed)
torch.manual_seed(4)
ref_mp = BondMessagePassing(activation=act, **dims)
G = torch.randn(bmg.V.shape[0], 300, generator=torch.Generator().manual_seed(6))


def err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def oracle(dtype):
    ps = [p.detach().to(dtype).requires_grad_(True) for p in (ref_mp.W_i.weight, ref_mp.W_h.weight, ref_mp.W_o.weight, ref_mp.W_o.bias)]
    out = ot.forward(bmg.V.to(dtype), bmg.E.to(dtype), bmg.edge_index, bmg.rev_edge_index, ot.MPWeights(*ps), depth=3, activation=act)
    (out * G.to(dtype)).sum().backward()
    return out.detach(), [p.grad for p in ps]


o64, g64 = oracle(torch.float64)
o32, g32 = oracle(torch.float32)
dev = torch.device("cuda:0")
mp = BondMessagePassing(activation=act, **dims)
mp.load_state_dict(ref_mp.state_dict())
mp = mp.to(dev).train()
bmg.to(dev)
out = mp(bmg)
(out * G.to(dev)).sum().backward()
env = {k: v for k, v in os.environ.items() if k.startswith("DMPNN_")}
print(f"{kind}-{n_mols} act={act} E={bmg.E.shape[0]} env={env}")
print(f"  out: engine vs fp64 {err(out.detach().cpu(), o64):.2e}   torch32 vs fp64 {err(o32, o64):.2e}")
got = [mp.W_i.weight.grad, mp.W_h.weight.grad, mp.W_o.weight.grad, mp.W_o.bias.grad]
for name, g, r32, r64 in zip(("W_i", "W_h", "W_o", "b_o"), got, g32, g64):
    g = g.cpu().numpy()
    d = np.abs(g - r64.numpy())
    print(f"  d{name}: engine vs fp64 {err(g, r64):.2e}  torch32 vs fp64 {err(r32, r64):.2e}  engine vs torch32 {err(g, r32):.2e}"
          f"  |max|={np.abs(r64.numpy()).max():.3g}  n(>1e-5 rel)={(d > 1e-5 * max(1, np.abs(r64.numpy()).max())).sum()} of {d.size}"
          f"  argmax={np.unravel_index(d.argmax(), d.shape)}")
