#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out/x; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -p no:cacheprovider -k "half_storage or per_step_fused or full_size or reference_class or wide" 2>&1 | grep -v "^  File\|^Extension modules" | tail -12 | cut -c1-300 | tee $OUT/pytest.txt
for xp in 0 1; do
DMPNN_XPATH=$xp timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench$xp.json 2> $OUT/bench.err
python - <<PY | tee -a $OUT/pytest.txt
import json
d=json.loads(open("gpurun_out/x/bench$xp.json").read().strip().splitlines()[-1])
print("XPATH=$xp value", d["value"])
for k,v in d.get("other_configs",{}).items(): print("  ", k, v.get("us"), v.get("f16_storage_us"))
PY
done
DMPNN_XPATH=1 python scripts/prof_large.py 2>/dev/null | tee -a $OUT/pytest.txt
DMPNN_XPATH=0 python scripts/prof_large.py 2>/dev/null | tee -a $OUT/pytest.txt
