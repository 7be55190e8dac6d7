#!/bin/bash
# prologue work of the tile kernels (kernel-argument warm-up, incidence fragments): phase stamps, parity, bench legs old vs new
export TMPDIR=/tmp
OUT=gpurun_out/r3warm; mkdir -p $OUT
timeout 120 ./scripts/micro/store_pattern 2>&1 | head -9 | tee $OUT/summary.txt
echo "== stamps" | tee -a $OUT/summary.txt; DMPNN_LIB=$PWD/chemprop_amd/variants/libdmpnn_metastamps.so python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | sed -n 2,26p | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -x -k "parity or atom or dropout or model or backward or train" 2>&1 | tail -5 | tee -a $OUT/summary.txt
for v in ${VARIANTS:-nowarm intree nowarm intree}; do
  if [ $v = intree ]; then unset DMPNN_LIB; else export DMPNN_LIB=$PWD/chemprop_amd/variants/libdmpnn_$v.so; fi
  echo "== $v" | tee -a $OUT/summary.txt; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('fwd us', d['ms_per_step']*1e3, 'value', d['value'], 'train', d.get('train_step', {}).get('ms_per_step'), 'model', d.get('model_step', {}).get('fused_ms_per_step'))" | tee -a $OUT/summary.txt
done
