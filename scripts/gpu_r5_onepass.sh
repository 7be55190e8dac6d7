#!/bin/bash
# round 5: the segment epilogue of k_step16 in one pass (dmpnn_seg16.hpp: seg_epilogue_1pass; the in-tree build) against the two-pass
# form (variant built with -DDMPNN_STEP16_TWOPASS: python scripts/build_variant.py twopass -DDMPNN_STEP16_TWOPASS).
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r5_onepass.sh <tag>'
TAG=${1:-r05_onepass}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
V=chemprop_amd/variants
{
echo "== parity of the per-step fused route, in-tree build (one pass)"
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_reference_class.py tests/test_spill.py tests/test_torch_export.py tests/test_model.py -q -m gpu -p no:cacheprovider -k "fused16 or fused_route or at_size or large or lean or store16 or per_step or real_subclass or oversize or spill or export or configs or golden or h0" 2>&1 | tail -12 | cut -c1-250
echo "== A/B configs 2-4: in-tree (one pass) | twopass | in-tree | twopass"
for i in 1 2; do
  echo "-- one pass"; timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids
  echo "-- two passes"; DMPNN_LIB=$V/libdmpnn_twopass.so timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids
done
echo "== stamps, in-tree"
DMPNN_STAMPS2=0 timeout 200 python scripts/probe_stamps_step16b.py 4096 synth40 2>&1 | grep -v amdgpu.ids | tail -11
} 2>&1 | tee $OUT/summary.txt
