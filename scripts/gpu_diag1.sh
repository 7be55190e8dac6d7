#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/diag1; mkdir -p $OUT; cd $REPO
{
python scripts/diag_grad.py cgr 512 relu
python scripts/diag_grad.py cgr 512 tanh
DMPNN_MFMA=f32 python scripts/diag_grad.py cgr 512 relu
DMPNN_BWD16=0 python scripts/diag_grad.py cgr 512 relu
python scripts/diag_grad.py cgr 128 relu
python scripts/diag_grad.py qm9 4096 relu
python scripts/diag_grad.py synth40 512 relu
} > $OUT/diag.txt 2>&1
cat $OUT/diag.txt | grep -v amdgpu.ids
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "spill or reference_class or collate or mab or whole_forward or caller_order" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest.log | cut -c1-300
