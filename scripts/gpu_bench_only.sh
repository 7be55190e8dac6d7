#!/bin/bash
# the two bench lines only (forward with side measurements; training step) -> gpurun_out/bench_only/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/bench_only; mkdir -p $OUT
timeout 900 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 100 --warmup 10 --mode train --no-cpu-baseline > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "train rc=$?"
python - <<'PY'
import json
for f in ("bench", "bench_train"):
    d = json.loads(open(f"gpurun_out/bench_only/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("train_step", {}).get("ms_per_step"), d["roofline"]["frac"], d["roofline"]["launch_us"])
PY
