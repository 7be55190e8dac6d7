#!/bin/bash
# round 5: H0 kept as row quads on the per-step fused route's inference forward (dmpnn_fwd_args.h0_bytes) against the forms of ABI <= 11
# (DMPNN_H0=x: the residual recomputed from the split K1 operand, d_h <= 320; fp32 rows read back word by word, d_h > 320).
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r5_h0q.sh <tag>'
TAG=${1:-r05_h0q}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
{
echo "== parity of the per-step fused route (H0 as row quads: the default)"
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_reference_class.py tests/test_spill.py tests/test_torch_export.py -q -m gpu -p no:cacheprovider -k "fused16 or fused_route or at_size or large or lean or store16 or per_step or real_subclass or oversize or spill or export or configs" 2>&1 | tail -12 | cut -c1-250
echo "== A/B configs 2-4: DMPNN_H0=x | quads | x | quads"
for m in x quads x quads; do echo "-- DMPNN_H0=$m"; DMPNN_H0=$m timeout 300 python scripts/ab_configs.py 2>&1 | grep -v amdgpu.ids; done
} 2>&1 | tee $OUT/summary.txt
