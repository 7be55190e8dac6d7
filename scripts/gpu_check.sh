#!/bin/bash
# One gpurun call: smoke, GPU parity tests, bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [tag]'
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
echo "== smoke" | tee $OUT/summary.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/smoke.log | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -40 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt; tail -5 $OUT/bench.err | tee -a $OUT/summary.txt
echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*stats*" | head; 
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -25 $f | tee -a $OUT/summary.txt; done
# keep the merge small
find $OUT/prof -name "*.db" -size +20M -delete; find $OUT/prof -name "*trace.csv" -size +20M -delete
echo "== done" | tee -a $OUT/summary.txt
