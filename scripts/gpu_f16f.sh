#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/f16f; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc -o p -- python $REPO/scripts/bench_configs.py $OUT/x.json synth40-4096 > $OUT/run.txt 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    if "k_step16" in k or "k_rows16<5, 4, true" in k or "k_rows16<5, 4, false" in k:
        k=k[:40]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
for k,v in agg.items():
    n=cnt[k]
    print(k, "launches", n)
    for c,x in v.items(): print("   %-28s %.4g per launch"%(c, x/n))
    if v.get("GRBM_GUI_ACTIVE"):
        print("   avg waves per CU = %.2f"%(v["SQ_WAVE_CYCLES"]*4/(v["GRBM_GUI_ACTIVE"]*256)))
        print("   MFMA busy / (4 SIMD x CU x active) = %.3f"%(v["SQ_VALU_MFMA_BUSY_CYCLES"]/(v["GRBM_GUI_ACTIVE"]*256*4)))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete; true
