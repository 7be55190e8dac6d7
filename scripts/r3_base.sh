#!/bin/bash
# round-3 baseline: GPU tests, bench as the driver runs it, host profile of the training step
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3base}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log | cut -c1-300 | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 rc=$?" | tee -a $OUT/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-large-batches > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
for f in ("bench20","bench"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "ms", d["ms_per_step"], "eager", d.get("eager_ms_per_step"), "graph", d.get("graph_ms_per_step"), "train", d.get("train_step",{}).get("ms_per_step"), "roof", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("launch_us"))
        for k,v in (d.get("other_configs") or {}).items(): print("   ", k, v)
    except Exception as e: print(f, "ERR", e)
PY
timeout 300 python scripts/probe_train_host2.py > $OUT/host_train.txt 2>&1; head -60 $OUT/host_train.txt | cut -c1-200 | tee -a $OUT/summary.txt
timeout 300 python scripts/host_profile.py > $OUT/host_fwd.txt 2>&1; head -40 $OUT/host_fwd.txt | cut -c1-200 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
