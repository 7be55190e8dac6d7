#!/bin/bash
# same-box A/B of variant builds on the training legs only (block step, fused model step)
export TMPDIR=/tmp
for v in intree "$@" intree "$@"; do
  if [ $v = intree ]; then unset DMPNN_LIB; else export DMPNN_LIB=$PWD/chemprop_amd/variants/libdmpnn_$v.so; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-large-batches 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$v', 'train', d.get('train_step', {}).get('ms_per_step'), 'model', d.get('model_step', {}).get('fused_ms_per_step'))"
done
