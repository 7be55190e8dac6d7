#!/bin/bash
# GPU tests (optionally a -k subset) + one default bench line -> gpurun_out/<tag>/
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_tests_bench.sh <tag> ["<pytest -k expression>"] [nobench]'
TAG=${1:-t}; KEXPR=${2:-}; NOBENCH=${3:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
if [ -n "$KEXPR" ]; then
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "$KEXPR" > $OUT/pytest.log 2>&1
else
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1
fi
echo "pytest rc=$?" | tee $OUT/summary.txt
grep -v "^  File\|^Extension modules" $OUT/pytest.log | tail -60 | cut -c1-300 | tee -a $OUT/summary.txt
if [ -z "$NOBENCH" ]; then
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    keys = ("value", "ms_per_step", "eager_ms_per_step", "graph_ms_per_step", "graph_k_steps_ms_per_step", "eager_ms_presplit_every_step", "best_launch_mode")
    print({k: d.get(k) for k in keys})
    print("roofline", {k: d["roofline"].get(k) for k in ("frac", "launch_us", "achieved")})
    print("train_step", d.get("train_step", {}).get("ms_per_step"), "model_step", {k: v for k, v in d.get("model_step", {}).items() if k.endswith("per_step") or k == "route" or k == "error"})
    for k, v in d.get("other_configs", {}).items():
        print(k, {a: b for a, b in v.items() if a in ("us", "route", "train_step_us", "train_route", "error", "f16_storage_us")})
    print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu"))
except Exception as e:
    print("bench parse failed:", e)
PY
  tail -3 $OUT/bench.err | cut -c1-300 | tee -a $OUT/summary.txt
fi
