#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/f16d; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in base x_NOMFMA x_NOSEG x_NOLOAD; do
  if [ $v != base ]; then export DMPNN_LIB=$REPO/chemprop_amd/variants/libdmpnn_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o p -- python $REPO/scripts/bench_configs.py $OUT/x.json synth40-4096 > $OUT/run_$v.txt 2>&1
  echo "== $v"; grep "now" $OUT/run_$v.txt
  for f in $(find $OUT/prof_$v -name "*kernel_stats.csv"); do grep "k_step16\|k_rows16<5, 4, true" $f | cut -c1-160; done
  python - <<PY
import csv,glob
f=glob.glob("$OUT/prof_$v/**/*kernel_trace.csv", recursive=True)
if f:
    rows=[r for r in csv.DictReader(open(f[0])) if "k_step16" in r["Kernel_Name"]]
    d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
    ev=d[0::2]; od=d[1::2]
    print("  k_step16 even launches (with Mout) avg %.1f us, odd (Mv) avg %.1f us"%(sum(ev)/len(ev)/1e3, sum(od)/len(od)/1e3))
PY
done
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete
