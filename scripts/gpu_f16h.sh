#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
python scripts/probe_step16.py cgr 512 2>&1 | grep -v amdgpu
python scripts/probe_step16.py synth40 4096 2>&1 | grep -v amdgpu
python scripts/probe_step16.py zinc 512 512 2>&1 | grep -v amdgpu
