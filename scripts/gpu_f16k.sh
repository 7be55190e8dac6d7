#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
for v in 0 1; do echo "== DMPNN_K1_SPLIT=$v"; DMPNN_K1_SPLIT=$v python scripts/bench_configs.py /tmp/x.json synth40 cgr-512 "zinc-512 h300" 2>&1 | grep "now"; done
