#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run12}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -9 | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_atom_mp.py tests/test_mab.py -q -m gpu -p no:cacheprovider -x > $OUT/pytest_atom.log 2>&1; echo "pytest(atom) rc=$?" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_atom.log | tail -25 | cut -c1-300 | tee -a $OUT/summary.txt
cat > /tmp/atom_time.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from chemprop_amd import synth
from chemprop_amd.nn import AtomMessagePassing, BondMessagePassing
dev = torch.device("cuda:0")
for n in (512, 4096):
    b = synth.random_batch(n, "qm9", seed=1000); b.to(dev)
    for cls in (BondMessagePassing, AtomMessagePassing):
        torch.manual_seed(0)
        m = cls().eval().to(dev)
        with torch.no_grad():
            for _ in range(10): m(b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): m(b)
            e1.record(); torch.cuda.synchronize()
        print(f"{cls.__name__} {n} mols: forward {e0.elapsed_time(e1) * 5:.1f} us  route={m.__dict__.get('_dmpnn_route')}")
PY
python /tmp/atom_time.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
