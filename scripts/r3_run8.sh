#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run8}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -8 | tee $OUT/summary.txt
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_gpu.log | tail -30 | cut -c1-300 | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.loads(open("$OUT/bench20.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "train", d.get("train_step",{}).get("ms_per_step"), "model", (d.get("model_step") or {}).get("fused_ms_per_step"), "roof", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("launch_us"))
print("cpu", {k: d.get("cpu_baseline",{}).get(k) for k in ("value","cores","kind")}, "cpu_train", {k: d.get("cpu_baseline_train",{}).get(k) for k in ("value","cores","kind")})
for k,v in (d.get("other_configs") or {}).items(): print("   ", k, v)
PY
echo "== done" | tee -a $OUT/summary.txt
