#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/f16e; mkdir -p $OUT; cd $REPO
timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "per_step_fused_route or full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
for st in 0 1 2 3; do echo "== DMPNN_STAGGER=$st"; export DMPNN_STAGGER=$st
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$st -o p -- python $REPO/scripts/bench_configs.py $OUT/x.json synth40-4096 > $OUT/run_$st.txt 2>&1
  grep "now" $OUT/run_$st.txt
  python - <<PY
import csv,glob
f=glob.glob("$OUT/prof_$st/**/*kernel_trace.csv", recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if "k_step16" in r["Kernel_Name"]]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
print("  k_step16 (with Mout) avg %.1f us, (Mv) avg %.1f us"%(sum(d[0::2])/max(len(d[0::2]),1)/1e3, sum(d[1::2])/max(len(d[1::2]),1)/1e3))
rows=[r for r in csv.DictReader(open(f[0])) if "k_rows16<5, 4, true" in r["Kernel_Name"]]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
print("  K1 seg avg %.1f us"%(sum(d)/len(d)/1e3))
PY
done
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete; true
