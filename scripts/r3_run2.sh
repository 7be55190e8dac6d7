#!/bin/bash
# round 3, second GPU call: full GPU tests on the two-workgroups-per-CU tile kernels, tile A/B, training step (in-tree vs non-temporal
# kept stores), bench line with the executed-reference CPU baseline, rocprofv3 of the training step
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
TAG=${1:-r3run2}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_gpu.log | tail -30 | cut -c1-300 | tee -a $OUT/summary.txt
timeout 200 python scripts/ab_tile.py 512 1024 4096 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
V=$REPO/chemprop_amd/variants/libdmpnn_ntkeep.so
for rep in 1 2; do
for lib in "" $V; do
DMPNN_LIB=$lib timeout 300 python bench.py --steps 100 --warmup 10 --mode train --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=[$lib] train step %.1f us  %.1f M/s'%(d['ms_per_step']*1e3, d['value']))" | tee -a $OUT/summary.txt
done; done
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.loads(open("$OUT/bench20.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["timing"], "graph", d.get("graph_ms_per_step"), "train", d.get("train_step",{}).get("ms_per_step"), "roof", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("launch_us"))
print("cpu", {k: d.get("cpu_baseline",{}).get(k) for k in ("value","cores","kind","ms_per_step")}, "cpu_train", {k: d.get("cpu_baseline_train",{}).get(k) for k in ("value","cores","kind","ms_per_step")}, d.get("cpu_baseline_reference_error"))
for k,v in (d.get("other_configs") or {}).items(): print("   ", k, v)
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $REPO/bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches > $OUT/prof_train.json 2> $OUT/prof_train.err
for f in $(find $OUT/prof_train -name "*kernel_stats.csv"); do head -14 $f | cut -c1-200 | tee -a $OUT/summary.txt; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
echo "== done" | tee -a $OUT/summary.txt
