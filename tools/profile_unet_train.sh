#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the UNet training step at 24^3 (tools/bench_train.py).  Output -> gpurun_out/prof_<tag>_unet_train/
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_unet_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py > $OUT/bench.txt 2> $OUT/err.txt
find $OUT -name "*kernel_stats.csv" | head -2
