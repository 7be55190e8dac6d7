import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chemprop_amd import synth, _lib
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bmg = synth.random_batch(n, "qm9", seed=13); bmg.to(dev)
torch.manual_seed(4)
mp = BondMessagePassing().to(dev).train()
print("fwd", flush=True)
out = mp(bmg); torch.cuda.synchronize()
print("fwd ok", out.grad_fn.st.route, flush=True)
(out * torch.randn_like(out)).sum().backward(); torch.cuda.synchronize()
print("bwd ok", float(mp.W_h.weight.grad.abs().max()), flush=True)
