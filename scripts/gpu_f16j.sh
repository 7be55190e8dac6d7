#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/f16j; mkdir -p $OUT; cd $REPO
python scripts/probe_step16.py synth40 4096 2>&1 | grep -v amdgpu
python scripts/probe_step16.py cgr 512 2>&1 | grep -v amdgpu
cd /tmp; export TMPDIR=/tmp
export DMPNN_FUSED16=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $REPO/scripts/bench_configs.py $OUT/x.json synth40-4096 > $OUT/run.txt 2>&1
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -16 $f | cut -c1-200; done
python - <<PY
import csv,glob
f=glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if "k_step16" in r["Kernel_Name"]]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
print("k_step16 (with Mout) avg %.1f us, (Mv) avg %.1f us"%(sum(d[0::2])/max(len(d[0::2]),1)/1e3, sum(d[1::2])/max(len(d[1::2]),1)/1e3))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete; true
