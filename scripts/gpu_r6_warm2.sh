#!/bin/bash
# round 6: where the L2 warm-up pays — module forward over batch sizes, warm-up always (variant warmall) | never (variant nowarm) | in-tree window
TAG=${1:-r06_warm_sizes}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=chemprop_amd/variants
{
for v in "" nowarm "" nowarm; do
  L=""; [ -n "$v" ] && L=$V/libdmpnn_$v.so
  DMPNN_LIB=$L timeout 300 python scripts/ab_tile.py 64 256 512 576 640 1024 2>&1 | grep -v amdgpu.ids | sed 's/tile kernel .* module forward/module forward/'
done
} 2>&1 | tee $OUT/summary.txt
