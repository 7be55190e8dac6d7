import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chemprop_amd import engine, synth
from chemprop_amd.nn import BondMessagePassing
dev = torch.device("cuda:0")
bmg = synth.random_batch(64, "qm9", seed=33); bmg.to(dev)
torch.manual_seed(8)
d_h, depth = 300, 3
mp = BondMessagePassing(d_h=d_h, depth=depth, activation="relu", bias=True).to(dev)
res = {}
for bits in (True, False):
    plan = engine.GraphPlan.from_bmg(bmg, light="tiles")
    out, st = engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, mp.W_i.bias, mp.W_h.bias,
                             depth=depth, act="relu", slope=0.0, keep=True, keep_bits=bits)
    torch.cuda.synchronize()
    res[bits] = (st, plan)
stb, planb = res[True]; stf, planf = res[False]
arr = planf.arrays()
nm = int(arr["hdr"][6]); mrow = arr["mtile_row"].numpy(); 
bitsbuf = stb.refs[-2].cpu().numpy().view(np.uint64)
import ctypes
from chemprop_amd import _lib
nb = stb.args.keep_bits_bytes
slot_words = nb // 8 // depth
print("tiles", nm, "slot words", slot_words, "max tiles", slot_words // 256)
H0 = stf.H0.cpu().numpy(); Hs = stf.Hs.cpu().numpy()
WN = 5
for slot, Hm in ((0, H0), (1, Hs[0]), (2, Hs[1])):
    bad = tot = 0
    for t in range(min(nm, 6)):
        rs, re = int(mrow[t]), int(mrow[t + 1])
        for w in range(4):
            for idx in range(60):
                word = int(bitsbuf[slot * slot_words + t * 256 + w * 64 + idx])
                r = idx & 3; ct = (idx >> 2) % WN; rt = (idx >> 2) // WN
                for l in range(64):
                    row = rt * 16 + (l >> 4) * 4 + r; col = (w * WN + ct) * 16 + (l & 15)
                    if row < re - rs and col < d_h:
                        tot += 1
                        if ((word >> l) & 1) != int(Hm[rs + row, col] > 0): bad += 1
    print("slot", slot, "mismatch", bad, "of", tot)
    if bad:
        import collections
        by = collections.Counter()
        t = 0; rs, re = int(mrow[0]), int(mrow[1])
        for w in range(4):
            for idx in range(60):
                word = int(bitsbuf[slot * slot_words + t * 256 + w * 64 + idx])
                r = idx & 3; ct = (idx >> 2) % WN; rt = (idx >> 2) // WN
                for l in range(64):
                    row = rt * 16 + (l >> 4) * 4 + r; col = (w * WN + ct) * 16 + (l & 15)
                    if row < re - rs and col < d_h and ((word >> l) & 1) != int(Hm[rs + row, col] > 0):
                        by[("w", w)] += 1; by[("rt", rt)] += 1; by[("ct", ct)] += 1; by[("r", r)] += 1; by[("lg", l >> 4)] += 1
        print(sorted(by.items(), key=str))
        # does the word match ANOTHER tensor's sign (H0)?
        for other, name in ((H0, "H0"), (Hs[0], "H1"), (Hs[1], "H2")):
            b2 = 0
            for w in range(4):
                for idx in range(60):
                    word = int(bitsbuf[slot * slot_words + w * 64 + idx])
                    r = idx & 3; ct = (idx >> 2) % WN; rt = (idx >> 2) // WN
                    for l in range(64):
                        row = rt * 16 + (l >> 4) * 4 + r; col = (w * WN + ct) * 16 + (l & 15)
                        if row < re - rs and col < d_h and ((word >> l) & 1) != int(other[rs + row, col] > 0): b2 += 1
            print("   tile 0 vs", name, "mismatch", b2)
