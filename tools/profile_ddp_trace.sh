#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of bench.py with only the configs[3] training leg beside a short headline.  Output -> gpurun_out/prof_<tag>_ddp/
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_ddp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-other-paths --no-roofline > $OUT/bench.txt 2> $OUT/err.txt
find $OUT -name "*kernel_trace.csv" | head -2
