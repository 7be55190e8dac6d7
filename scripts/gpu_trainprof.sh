#!/bin/bash
# rocprofv3 kernel stats of the training step (bench.py --mode train) -> gpurun_out/trainprof/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/trainprof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o train -- python $REPO/bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-large-batches > $OUT/train.json 2> $OUT/err.txt
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -14 $f | cut -c1-170; done
python - <<'PY'
import csv, glob, collections
f = glob.glob("/root/repo/gpurun_out/trainprof/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    if "k_wgrad16" in r["Kernel_Name"] or "k_wsplit16" in r["Kernel_Name"] or "k_wgrad_reduce" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:40], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()): print(k, len(v), "avg us", round(sum(v) / len(v), 1))
PY
