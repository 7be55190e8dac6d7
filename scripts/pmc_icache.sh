#!/bin/bash
# instruction-cache / issue counters of the forward kernels (one rocprofv3 --pmc pass per counter group)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  d=$OUT/$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $d -o pmc --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $d.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc_icache/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in agg.items():
    if "dmpnn" not in k and "mpnn" not in k and "prepare" not in k: continue
    print(k)
    for c, (v, n) in sorted(cs.items()):
        print(f"   {c:28s} {v / n:14.0f}  (n={n})")
PY
