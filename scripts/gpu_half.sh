#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/half; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -p no:cacheprovider -k "half_storage or per_step_fused" -s 2>&1 | grep -v "^  File\|^Extension modules" | grep "half storage\|passed\|failed\|Error\|assert" | tail -25 | cut -c1-300 | tee $OUT/pytest.txt
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY' | tee -a $OUT/pytest.txt
import json
d=json.loads(open("gpurun_out/half/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "train_step", d.get("train_step"))
for k,v in d.get("other_configs",{}).items(): print(k, v)
PY
tail -3 $OUT/bench.err
