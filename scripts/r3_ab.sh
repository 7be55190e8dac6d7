#!/bin/bash
# same-box A/B of a variant build (chemprop_amd/variants/libdmpnn_$1.so) against the in-tree library: stamps, parity subset, bench legs
export TMPDIR=/tmp
V=$1; OUT=gpurun_out/ab_$V; mkdir -p $OUT
echo "== stamps $V" | tee $OUT/summary.txt; DMPNN_LIB=$PWD/chemprop_amd/variants/libdmpnn_$V.so python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | sed -n 2,22p | tee -a $OUT/summary.txt
DMPNN_LIB=$PWD/chemprop_amd/variants/libdmpnn_$V.so timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_atom_mp.py -q -m gpu -x -k "${2:-forward or golden or tile or atom or backward}" 2>&1 | tail -3 | tee -a $OUT/summary.txt
for v in intree $V intree $V; do
  if [ $v = intree ]; then unset DMPNN_LIB; else export DMPNN_LIB=$PWD/chemprop_amd/variants/libdmpnn_$v.so; fi
  echo "== $v" | tee -a $OUT/summary.txt; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('fwd us', d['ms_per_step']*1e3, 'value', d['value'], 'kernel us', d['roofline'].get('launch_us'), 'train', d.get('train_step', {}).get('ms_per_step'), 'model', d.get('model_step', {}).get('fused_ms_per_step'))" | tee -a $OUT/summary.txt
done
