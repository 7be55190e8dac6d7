import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chemprop_amd import synth
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
os.environ["DMPNN_VALIDATE"] = "never"
n_mols, p = 2048, 0.3
bmg = synth.random_batch(n_mols, "qm9", seed=33); bmg.to(dev)
torch.manual_seed(4)
mp = BondMessagePassing(dropout=p).to(dev).train()
G = torch.randn(bmg.V.shape[0], 300, generator=torch.Generator().manual_seed(6)).to(dev)
res = {}
for kind in (sys.argv[1:] or ["tiles", "full"]):
    os.environ["DMPNN_TRAIN_PLAN"] = kind
    mp.zero_grad(); torch.manual_seed(1234)
    out = mp(bmg); st = out.grad_fn.st
    print(kind, "route", st.route, "tiles_only", st.plan.tiles_only, "flags", hex(st.args.flags), "seed", st.args.dropout_seed, flush=True)
    H0, Hs, Ms, Mv = st.H0.clone(), st.Hs.clone(), st.Ms.clone(), st.Mv.clone()
    perm = None if st.plan.tiles_only else st.plan.perm64.clone()
    (out * G).sum().backward()
    kind = kind + ("2" if kind in res else "")
    res[kind] = dict(out=out.detach().clone(), H0=H0, Hs=Hs, Ms=Ms, Mv=Mv, perm=perm, g={k: q.grad.clone() for k, q in mp.named_parameters()})
for k in res: print(k, "gb_o sum", float(res[k]["g"]["W_o.bias"].double().sum()), "gW_o norm", float(res[k]["g"]["W_o.weight"].double().norm()), "gW_i norm", float(res[k]["g"]["W_i.weight"].double().norm()))
if "tiles" not in res or "full" not in res: sys.exit(0)
t, f = res["tiles"], res["full"]
inv = torch.empty_like(f["perm"]); inv[f["perm"]] = torch.arange(f["perm"].numel(), device=dev)
def err(a, b): return float((a - b).abs().max() / b.abs().max())
print("out", err(f["out"], t["out"]), "Mv", err(f["Mv"], t["Mv"]))
print("H0", err(f["H0"][inv], t["H0"]), "Hs0", err(f["Hs"][0][inv], t["Hs"][0]), "Hs1", err(f["Hs"][1][inv], t["Hs"][1]),
      "Ms0", err(f["Ms"][0][inv], t["Ms"][0]), "Ms1", err(f["Ms"][1][inv], t["Ms"][1]))
bad = ((f["Hs"][1][inv] - t["Hs"][1]).abs().amax(dim=1) > 1e-4).nonzero().flatten()
print("rows of Hs1 that differ:", bad.numel(), bad[:10].tolist(), "of", t["Hs"].shape[1])
for k in t["g"]: print(k, err(f["g"][k], t["g"][k]))
