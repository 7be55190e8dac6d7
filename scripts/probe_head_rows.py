"""Round 5: the whole-model training step at 512 QM9-shaped molecules with the head in the row / column kernels (default) against the nine-launch
chain (DMPNN_HEAD=chain, read per call): wall time per step of FusedTrainer.step and of the module path, alternating; the losses of 20
steps from the same start under both forms.  With `prof` as argument: 200 fused steps only (the target of rocprofv3 --kernel-trace)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chemprop_amd import agg as cagg, distributed as ddp, synth
from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN
from chemprop_amd.nn import BondMessagePassing
from chemprop_amd.optim import FlatAdam

dev = torch.device("cuda:0")
n_mols = int(os.environ.get("MOLS", "512"))
b = synth.random_batch(n_mols, "qm9", seed=1000); b.to(dev)
y = torch.randn(n_mols, 1, device=dev)


def model():
    torch.manual_seed(0)
    return MPNN(BondMessagePassing(d_h=300), cagg.NormAggregation(), RegressionFFN(n_tasks=1, input_dim=300), batch_norm=True).to(dev).train()


if len(sys.argv) > 1 and sys.argv[1] == "prof":
    tr = FusedTrainer(model(), lr=1e-4)
    for _ in range(200): tr.step(b, y)
    torch.cuda.synchronize()
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "stamps":   # cycle stamps of one workgroup of each of the head's three kernels (a 128-entry buffer)
    from chemprop_amd import _lib
    lib = _lib.load()
    tr = FusedTrainer(model(), lr=1e-4)
    for _ in range(5): tr.step(b, y)
    buf = torch.zeros(128, dtype=torch.int64, device=dev)
    names = {64: ("k_head_rows<.., 1>, workgroup (1, 1)", ["entry", "requests out", "contraction issued", "end"]),
             72: ("k_head_rows<.., 2>, workgroup (1, 1)", ["entry", "requests out", "A1 in the tile", "criterion", "output layer's backward", "contraction issued", "end"]),
             80: ("k_agg_bn_fwd, column workgroup 1", ["entry", "requests out", "bounds here", "aggregated", "statistics", "end"]),
             86: ("k_agg_bn_fwd, first split workgroup", ["entry", "end"]),
             96: ("k_bn_agg_bwd, column workgroup 1", ["entry", "requests out", "data here", "column sums", "rows issued"])}
    for rep in range(2):
        buf.zero_()
        lib.dmpnn_debug_timestamps(buf.data_ptr())
        tr.step(b, y)
        torch.cuda.synchronize()
        lib.dmpnn_debug_timestamps(None)
        st = buf.cpu().tolist()
        for base, (title, nm) in names.items():
            print(f"--- rep {rep}: {title}")
            prev = st[base]
            for i, n in enumerate(nm):
                if st[base + i]:
                    print(f"    {n:32s} +{st[base + i] - prev:7d} cycles   (t = {st[base + i] - st[base]})")
                    prev = st[base + i]
    sys.exit(0)

losses, g1 = {}, {}
for form in ("rows", "chain"):
    os.environ["DMPNN_HEAD"] = form
    mm = model()
    tr = FusedTrainer(mm, lr=1e-3)
    losses[form] = [float(tr.step(b, y)[0])]
    names = [k for k, p in mm.named_parameters() if p.requires_grad]
    g1[form] = {k: tr.sync.views[i].detach().cpu().clone() for i, k in enumerate(names)}   # (the gradients of step 1)
    losses[form] += [float(tr.step(b, y)[0]) for _ in range(19)]
for k in g1["rows"]:
    a, c = g1["rows"][k], g1["chain"][k]
    print(f"  step-1 gradient {k:34s} max|rows - chain| / max|chain| = {float((a - c).abs().max() / c.abs().max().clamp_min(1e-30)):.2e}")
print("losses, the row / column kernels:", " ".join(f"{v:.6f}" for v in losses["rows"][::4]))
print("losses, chain         :", " ".join(f"{v:.6f}" for v in losses["chain"][::4]))
print("max relative difference over 20 steps:", max(abs(a - c) / max(abs(c), 1e-12) for a, c in zip(losses["rows"], losses["chain"])))

m = model(); tr = FusedTrainer(m, lr=1e-4)
m2 = model()
sync = ddp.GradSync([p for p in m2.parameters() if p.requires_grad], modules=[m2.message_passing])
opt = FlatAdam(sync, lr=1e-4)


def fused(): tr.step(b, y)


def module():
    with ddp.backward_on_calling_thread():
        sync.zero_grad()
        m2.loss(b, y).backward()
    sync.allreduce(); opt.step()


FORMS = [("chain", {"DMPNN_HEAD": "chain"}), ("default", {}), ("agg-fused", {"DMPNN_HEAD_AGG": "fused"}), ("agg-split", {"DMPNN_HEAD_AGG": "split"})]
for rep in range(2):
    for form, env in FORMS:
        for k in ("DMPNN_HEAD", "DMPNN_HEAD_AGG", "DMPNN_HEAD_QPW"): os.environ.pop(k, None)
        os.environ.update(env)
        for name, fn in (("fused", fused), ("module", module)):
            if name == "module" and form in ("agg-split", "agg-fused"): continue
            for _ in range(30): fn()
            torch.cuda.synchronize()
            n = 300
            t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"[{form:9s}] {name:7s}: {1e6 * (t2 - t0) / n:7.1f} us/step")
