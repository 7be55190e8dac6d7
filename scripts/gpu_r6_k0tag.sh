#!/bin/bash
# round 6 (late): K0's offsets as tagged 8-byte words the packing block polls directly (in-tree) against the flag-then-data hand-off (variant k0old)
TAG=${1:-r06_k0tag}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=chemprop_amd/variants
{
timeout 1200 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -3 | cut -c1-220
for v in "" k0old "" k0old; do
  L=""; [ -n "$v" ] && L=$V/libdmpnn_$v.so
  DMPNN_LIB=$L timeout 300 python scripts/ab_tile.py 64 256 512 1024 2>&1 | grep -v amdgpu.ids | sed 's/tile kernel [^|]*| //'
  echo "[${v:-in-tree}] $(DMPNN_LIB=$L python scripts/prof_model_step.py 512 400 2>&1 | grep -v amdgpu.ids | tail -1)"
done
for v in "" k0old; do
  L=""; [ -n "$v" ] && L=$V/libdmpnn_$v.so
  echo "--- packing block stamps, ${v:-in-tree}"
  for i in 1 2; do DMPNN_LIB=$L python scripts/probe_k0.py 512 2>&1 | grep -v amdgpu | tr '\n' ';'; echo; done
done
} 2>&1 | tee $OUT/summary.txt
