#!/bin/bash
# round 6: the finalize's V rows staged with the aggregate (in-tree) against the staged form (-DDMPNN_V_LATE, variant vlate), same box
TAG=${1:-r06_vearly}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=chemprop_amd/variants
{
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_model.py tests/test_dropout_gpu.py tests/test_atom_mp.py tests/test_mab.py tests/test_reference_class.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -3 | cut -c1-220
for st in f32 f16; do
for v in "" vlate "" vlate; do
  L=""; [ -n "$v" ] && L=$V/libdmpnn_$v.so
  DMPNN_STORE=$st DMPNN_LIB=$L timeout 300 python scripts/ab_tile.py 64 512 2>&1 | grep -v amdgpu.ids | sed 's/| K0 .* module forward/| module forward/' | sed "s/^/[$st] /"
done
done
for v in "" vlate "" vlate; do
  L=""; [ -n "$v" ] && L=$V/libdmpnn_$v.so
  echo "[${v:-in-tree}] $(DMPNN_LIB=$L python scripts/prof_model_step.py 512 400 2>&1 | grep -v amdgpu.ids | tail -1)"
done
echo "--- phase stamps, in-tree"
DMPNN_TILE_WAVES=8 timeout 100 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | head -23
} 2>&1 | tee $OUT/summary.txt
