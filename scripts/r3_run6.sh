#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run6}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -8 | tee $OUT/summary.txt
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_gpu.log | tail -12 | cut -c1-300 | tee -a $OUT/summary.txt
timeout 300 python scripts/host_profile.py 2>&1 | grep -v amdgpu.ids | head -14 | cut -c1-200 | tee -a $OUT/summary.txt
for st in 20 200; do
timeout 600 python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-large-batches > $OUT/bench$st.json 2> $OUT/bench$st.err; echo "bench$st rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.loads(open("$OUT/bench$st.json").read().strip().splitlines()[-1])
print("steps $st: value", d["value"], "ms", d["ms_per_step"], d["timing"]["eager_ms_per_step_by_group"], "graph", d.get("graph_ms_per_step"), "train", d.get("train_step",{}).get("ms_per_step"), "model", (d.get("model_step") or {}).get("fused_ms_per_step"), "roof", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("launch_us"))
PY
done
echo "== done" | tee -a $OUT/summary.txt
