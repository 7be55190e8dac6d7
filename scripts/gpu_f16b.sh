#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/f16b; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for cfg in synth40-4096 cgr-512 "zinc-512 h300"; do
tag=$(echo $cfg | tr ' ' '_')
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o p -- python $REPO/scripts/bench_configs.py $OUT/x.json "$cfg" > $OUT/run_$tag.txt 2>&1
echo "== $cfg"; grep -v amdgpu $OUT/run_$tag.txt | tail -2
for f in $(find $OUT/prof_$tag -name "*kernel_stats.csv"); do head -14 $f | cut -c1-220; done
done
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -size +5M -delete
