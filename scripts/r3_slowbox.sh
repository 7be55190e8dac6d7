#!/bin/bash
# what a "slow box" is slow at: HBM rates, then the phase stamps of the tile kernel (workgroup 0) as inference and as training forward
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/${1:-slowbox}; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python - <<PY 2>&1 | grep -v amdgpu.ids | tee $OUT/summary.txt
import torch
dev = torch.device("cuda:0")
a = torch.empty(256 << 20, dtype=torch.float32, device=dev); b = torch.empty_like(a)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
tc = t(lambda: b.copy_(a)); tf = t(lambda: a.fill_(1.0)); tr = t(lambda: a.sum())
print(f"HBM 1 GiB: copy {2 * a.numel() * 4 / tc / 1e12:.2f} TB/s (r+w)   fill {a.numel() * 4 / tf / 1e12:.2f} TB/s (w)   sum {a.numel() * 4 / tr / 1e12:.2f} TB/s (r)")
PY
for mode in "tiles" "tiles keep"; do echo "== stamps: $mode" | tee -a $OUT/summary.txt; python scripts/probe_stamps.py 512 $mode 2>&1 | grep -v amdgpu.ids | sed -n 2,22p | tee -a $OUT/summary.txt; done
python scripts/probe_model_host.py 2>&1 | grep -v amdgpu.ids | sed -n 1,1p | tee -a $OUT/summary.txt
