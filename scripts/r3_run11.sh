#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run11}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -9 | tee $OUT/summary.txt
timeout 120 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | head -30 | tee -a $OUT/summary.txt
timeout 120 python scripts/probe_stamps.py 512 keep 2>&1 | grep -v amdgpu.ids | head -30 | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "train", d.get("train_step",{}).get("ms_per_step"), "model", (d.get("model_step") or {}).get("fused_ms_per_step"), "dropout", d.get("train_step_dropout"), "roof", d.get("roofline",{}).get("frac"))
PY
tail -3 $OUT/bench.err | cut -c1-200 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
