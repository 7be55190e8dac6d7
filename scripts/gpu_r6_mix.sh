#!/bin/bash
# round 6: the pair split on v_fma_mix*_f16 + ReLU as v_maximum3_f32 (in-tree) against the C++ forms (variant cxxsplit), same box:
# bit-identity of the forward output, parity tests, tile kernel / configs 2-4 / whole-model step timings, phase stamps
TAG=${1:-r06_mix}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=chemprop_amd/variants
{

timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_model.py tests/test_dropout_gpu.py tests/test_atom_mp.py tests/test_mab.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -3 | cut -c1-220
for v in "" reluselect "" reluselect; do
  echo "--- variant '${v:-in-tree}'"
  L=""; [ -n "$v" ] && L=$V/libdmpnn_$v.so
  DMPNN_LIB=$L DMPNN_TILE_WAVES=8 timeout 200 python scripts/ab_tile.py 512 2>&1 | grep -v amdgpu.ids
  DMPNN_LIB=$L python scripts/prof_model_step.py 512 400 2>&1 | grep -v amdgpu.ids | tail -1
  DMPNN_LIB=$L timeout 300 python scripts/bench_configs.py /dev/null h512 synth40 cgr-512 2>&1 | grep " now "
done
echo "--- phase stamps, in-tree"
DMPNN_TILE_WAVES=8 timeout 100 python scripts/probe_stamps.py 512 tiles 2>&1 | grep -v amdgpu.ids | head -23
} 2>&1 | tee $OUT/summary.txt
