#!/bin/bash
# One gpurun call: smoke, GPU parity tests, bench, probe, rocprofv3 kernel stats + PMC traffic passes.
# Everything lands in gpurun_out/<tag>/.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [tag] [full|quick|prof]'   (prof: only the profiler passes)
TAG=${1:-r01}
QUICK=${2:-full}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
if [ "$QUICK" != "prof" ]; then
echo "== smoke" | tee $OUT/summary.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -6 $OUT/smoke.log | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; RC=$?; echo "pytest rc=$RC" | tee -a $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest_gpu.log | tail -40 | cut -c1-400 | tee -a $OUT/summary.txt
if [ $RC -ne 0 ] && [ $RC -ne 1 ]; then
  echo "== pytest crashed: traced re-run (kernel-by-kernel sync)" | tee -a $OUT/summary.txt
  DMPNN_TRACE=1 timeout 600 python -m pytest tests -q -x -m gpu -p no:cacheprovider -s > $OUT/pytest_trace.log 2>&1
  grep -v "^  File\|^Extension modules" $OUT/pytest_trace.log | grep -B2 -A12 "fault\|Abort\|error\|FAILED" | head -60 | cut -c1-300 | tee -a $OUT/summary.txt
  grep "\[dmpnn\] launch" $OUT/pytest_trace.log | tail -5 | tee -a $OUT/summary.txt
fi
echo "== probe" | tee -a $OUT/summary.txt
timeout 300 python scripts/gemm_probe.py > $OUT/probe.txt 2>&1; cat $OUT/probe.txt | tail -30 | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt; tail -5 $OUT/bench.err | tee -a $OUT/summary.txt
if [ "$QUICK" != "quick" ]; then
echo "== bench train" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 100 --warmup 10 --mode train --no-cpu-baseline > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train rc=$?" | tee -a $OUT/summary.txt
cut -c1-900 $OUT/bench_train.json | tee -a $OUT/summary.txt; tail -3 $OUT/bench_train.err | tee -a $OUT/summary.txt
fi
fi
echo "== rocprofv3 kernel stats (bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches)" | tee -a $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof rc=$?" | tee -a $OUT/summary.txt
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -16 $f | cut -c1-260 | tee -a $OUT/summary.txt; done
if [ "$QUICK" != "quick" ]; then
echo "== rocprofv3 kernel stats, training step (bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches)" | tee -a $OUT/summary.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $REPO/bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches > $OUT/prof_train.json 2> $OUT/prof_train.err
for f in $(find $OUT/prof_train -name "*kernel_stats.csv"); do head -12 $f | cut -c1-200 | tee -a $OUT/summary.txt; done
echo "== rocprofv3 PMC passes (separate runs, counters only)" | tee -a $OUT/summary.txt
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-large-batches > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-large-batches > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $OUT/pmc_sq -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-large-batches > $OUT/pmc_sq.log 2>&1
cd $REPO
python scripts/pmc_traffic.py $OUT | tee -a $OUT/summary.txt
echo "== 32 768 molecules (working set beyond the Infinity Cache): rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of scripts/prof_large.py" | tee -a $OUT/summary.txt
mkdir -p $OUT/large
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/large/prof -o large -- python $REPO/scripts/prof_large.py > $OUT/large/prof_large.json 2> $OUT/large/prof.err
cat $OUT/large/prof_large.json | tee -a $OUT/summary.txt
for f in $(find $OUT/large/prof -name "*kernel_stats.csv"); do head -8 $f | cut -c1-200 | tee -a $OUT/summary.txt; done
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/large/pmc_fetch -o p -- python $REPO/scripts/prof_large.py > $OUT/large/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/large/pmc_write -o p -- python $REPO/scripts/prof_large.py > $OUT/large/pmc_write.log 2>&1
cd $REPO
python scripts/pmc_traffic.py $OUT/large | tee -a $OUT/summary.txt
fi
# keep the merge small
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete
echo "== done" | tee -a $OUT/summary.txt
