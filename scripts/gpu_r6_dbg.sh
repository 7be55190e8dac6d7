#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
mkdir -p gpurun_out/r06_dbg
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
{
for v in bwdnohoist; do for w in 4 8; do
echo "=== variant $v waves $w"
DMPNN_LIB=chemprop_amd/variants/libdmpnn_$v.so DMPNN_TILE_WAVES=$w timeout 300 python scripts/dbg_bwd.py 512 2>&1 | grep -v amdgpu.ids | head -6 | cut -c1-250
done; done
} 2>&1 | tee gpurun_out/r06_dbg/summary.txt
