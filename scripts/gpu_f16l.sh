#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for pad in 0 40; do echo "== DMPNN_STEP16_PAD_LDS=$pad (KB)"; export DMPNN_STEP16_PAD_LDS=$pad
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$pad -o p -- python $REPO/scripts/bench_configs.py /tmp/x.json synth40-4096 > /tmp/run_$pad.txt 2>&1
grep now /tmp/run_$pad.txt
python - <<PY
import csv,glob
f=glob.glob("/tmp/prof_$pad/**/*kernel_trace.csv", recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if "k_step16" in r["Kernel_Name"]]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
print("k_step16 (with Mout) avg %.1f us, (Mv) avg %.1f us"%(sum(d[0::2])/max(len(d[0::2]),1)/1e3, sum(d[1::2])/max(len(d[1::2]),1)/1e3))
PY
done
