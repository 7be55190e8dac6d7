#!/bin/bash
# L2 -> CU traffic of the whole-forward tile kernel on the headline batch: every tile streams the whole split weights
# (W_i, 2 x W_h, W_o: ~1.6 MB) from its XCD's L2, so a launch of ~250 tiles moves ~400 MB through the L2s although it
# reads 8 MB from HBM.  Counters: vector-L1 -> L2 read requests, L2 requests / hits / misses (their own pass, kernel
# trace only).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/pmc_l2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/a -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-large-batches > $OUT/a.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/b -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-large-batches > $OUT/b.log 2>&1
python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_l2"
for sub in ("a", "b"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_mpnn_tile16" in k:
                acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print(f"  {c:36s} mean {sum(v)/len(v):14.1f}  over {len(v)} launches")
PY
tail -3 $OUT/a.log $OUT/b.log | cut -c1-300
