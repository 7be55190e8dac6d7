#!/bin/bash
# round 5: bench.py under torchrun at world size 1 on a REAL nccl (= RCCL) process group, the gradient exchange forced through its
# collective branch (DMPNN_FORCE_COLLECTIVE=1): the barriers, the MAX all-reduce of the timing and the flat gradient all-reduce of the
# training step all run on RCCL — the first execution of that code is then not the driver's 8-GPU run.
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_r5_nccl.sh <tag>'
TAG=${1:-r05_nccl}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
DMPNN_FORCE_COLLECTIVE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-large-batches > $OUT/bench_nccl_world1.json 2> $OUT/bench_nccl_world1.err
echo "rc=$?" | tee $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d = json.loads(open("$OUT/bench_nccl_world1.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus", "rccl")})
print("train_step", {k: d.get("train_step", {}).get(k) for k in ("ms_per_step", "collective", "collectives_launched", "allreduce_exposed_us", "error")})
PY
tail -3 $OUT/bench_nccl_world1.err | cut -c1-300 | tee -a $OUT/summary.txt
