import sys, os
sys.path.insert(0, os.getcwd())
import torch
from chemprop_amd import engine, synth, _lib
dev = torch.device("cuda:0"); lib = _lib.load()
bmg = synth.random_batch(512, "qm9", seed=1000); bmg.to(dev)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
for _ in range(5): engine.GraphPlan.from_bmg(bmg, light="tiles")
for rep in range(3):
    buf.zero_()
    lib.dmpnn_debug_timestamps(buf.data_ptr())
    engine.GraphPlan.from_bmg(bmg, light="tiles"); torch.cuda.synchronize()
    lib.dmpnn_debug_timestamps(None)
    st = buf.cpu().tolist(); t0 = st[32]
    print("pack:", [st[32+i]-t0 for i in range(5)], "pieces:", [st[48+i]-t0 if st[48+i] else 0 for i in range(3, 10)])
