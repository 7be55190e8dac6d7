#!/bin/bash
# dry run of the N = 2 control flow of bench.py on a ONE-GPU box (gloo, both ranks on cuda:0): does every rank reach every collective?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
mkdir -p gpurun_out/dry2
export DMPNN_BENCH_BACKEND=gloo DMPNN_BENCH_DEVICE=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/dry2/out.json 2> gpurun_out/dry2/err.txt
echo "rc=$?"
python - <<'PY'
import json
t = open("gpurun_out/dry2/out.json").read().strip().splitlines()
print(len(t), "line(s)")
d = json.loads(t[-1])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, d.get("train_step"))
PY
tail -5 gpurun_out/dry2/err.txt | cut -c1-200
