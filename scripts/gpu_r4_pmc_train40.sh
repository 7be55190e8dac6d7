#!/bin/bash
# PMC passes (counters only, separate runs) of the training step of 40-atom molecules x 4096: HBM-side bytes and SQ counters per kernel
TAG=${1:-r4pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
CMD="python $REPO/bench.py --mode train --kind synth40 --mols 4096 --steps 6 --warmup 3 --no-cpu-baseline --no-graph --no-large-batches"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
cd $REPO
python scripts/pmc_traffic.py $OUT | grep -A1 "bstep\|wgrad16t\|rows2blk\|k_step16\|k_rows16\|k_split_rows\|wsplit" | cut -c1-330 | tee $OUT/summary.txt
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*trace.csv" -size +30M -delete; find $OUT -name "*counter_collection.csv" -size +30M -delete
