#!/bin/bash
# round 5: the head of the training step in four launches (k_agg_bn_fwd / k_head_rows<., 1 | 2> / k_bn_agg_bwd) — its tests, the A/B against the
# chain (DMPNN_HEAD=chain), kernel stats of 200 fused steps.
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r5_head.sh <tag>'
TAG=${1:-r05_head}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
{
echo "== tests of the head and of the whole-model step"
timeout 900 python -m pytest tests/test_model.py tests/test_reference_class.py tests/test_lightning_fit.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension modules" | tail -40 | cut -c1-300
echo "== A/B: DMPNN_HEAD=chain | default | aggregation in front | 4 quads per column workgroup"
timeout 300 python scripts/probe_head_rows.py 2>&1 | grep -v amdgpu.ids
MOLS=64 timeout 300 python scripts/probe_head_rows.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== kernel stats of 200 fused steps"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o step -- python $REPO/scripts/probe_head_rows.py prof > /dev/null 2>&1
cd $REPO
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/model_step_kernel_stats.csv && cut -c1-150 $f | head -32
rm -rf $OUT/prof
} 2>&1 | tee $OUT/summary.txt
