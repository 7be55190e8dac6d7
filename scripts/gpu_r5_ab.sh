#!/bin/bash
# round 5: A/B of a k_step16 variant on BASELINE configs 2-4 + the parity tests of the per-step fused route on it.
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r5_ab.sh <tag> variant ...'
TAG=${1:-r05_ab}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=chemprop_amd/variants
{
for v in "$@"; do
  echo "== parity of the per-step fused route on $v"
  DMPNN_LIB=$V/libdmpnn_$v.so timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_reference_class.py tests/test_spill.py -q -m gpu -p no:cacheprovider -x -k "fused16 or fused_route or at_size or large or lean or store16 or per_step or real_subclass or oversize or spill" 2>&1 | tail -15 | cut -c1-250
done
echo "== A/B configs 2-4 (in-tree, variants, in-tree again)"
timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do DMPNN_LIB=$V/libdmpnn_$v.so timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids; done
timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  echo "== stamps, $v"
  DMPNN_STAMPS2=0 DMPNN_LIB=$V/libdmpnn_$v.so timeout 200 python scripts/probe_stamps_step16b.py 4096 synth40 2>&1 | grep -v amdgpu.ids | tail -11
done
} 2>&1 | tee $OUT/summary.txt
