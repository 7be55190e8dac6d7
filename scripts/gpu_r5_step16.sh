#!/bin/bash
# round 5: what the per-step fused kernel (k_step16) waits for between entry and its MFMA loop, and the XFIRST variant
# (the x path's requests ahead of the operand tile's DMA).  Variants are built HERE (scripts/build_variant.py) and travel in-tree.
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r5_step16.sh <tag> [variant ...]'
TAG=${1:-r05_s16}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=chemprop_amd/variants
{
echo "== stamps, measurement build (forced wait after the requests)"
DMPNN_LIB=$V/libdmpnn_s16stamps.so timeout 200 python scripts/probe_stamps_step16b.py 4096 synth40 2>&1 | grep -v amdgpu.ids
echo "== stamps, in-tree build"
DMPNN_STAMPS2=0 timeout 200 python scripts/probe_stamps_step16b.py 4096 synth40 2>&1 | grep -v amdgpu.ids | tail -12
for v in "$@"; do
  echo "== stamps, $v"
  DMPNN_STAMPS2=0 DMPNN_LIB=$V/libdmpnn_$v.so timeout 200 python scripts/probe_stamps_step16b.py 4096 synth40 2>&1 | grep -v amdgpu.ids | tail -12
done
echo "== A/B configs 2-4 (in-tree, variants, in-tree again)"
timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do DMPNN_LIB=$V/libdmpnn_$v.so timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids; done
timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  echo "== parity of the per-step fused route on $v"
  DMPNN_LIB=$V/libdmpnn_$v.so timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -p no:cacheprovider -k "fused16 or fused_route or at_size or large or lean or store16 or per_step" 2>&1 | tail -4
done
} 2>&1 | tee $OUT/summary.txt
