#!/bin/bash
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_probe.sh tag'
TAG=${1:-probe}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 300 python scripts/gemm_probe.py > $OUT/probe.txt 2>&1
cat $OUT/probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $OUT/pmc1 -o p -- python $REPO/scripts/gemm_probe.py one > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES -d $OUT/pmc2 -o p -- python $REPO/scripts/gemm_probe.py one > $OUT/pmc2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("pmc1", "pmc2"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, v in agg.items():
            if "k_linear" in k or "k_segment" in k:
                print(d, k, {c: round(x) for c, x in v.items()})
PY
