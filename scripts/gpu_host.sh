#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
mkdir -p gpurun_out/host
python scripts/host_profile.py 16 2>&1 | grep -v amdgpu.ids | head -50 | cut -c1-160 | tee gpurun_out/host/profile16.txt
python scripts/host_profile.py 512 2>&1 | grep "host issue" | tee -a gpurun_out/host/profile16.txt
