#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the AutoencoderKL 96^3 encode / decode (tools/bench_vae.py) plus one LDS counter pass
# (separate run, --kernel-trace only).  Output -> gpurun_out/prof_<tag>_vae/
set -u
TAG=${1:-r04c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_vae
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/bench_vae.py
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B > $OUT/bench_trace.txt 2> $OUT/trace.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/pmc_lds -- python3 $B > $OUT/bench_pmc.txt 2> $OUT/pmc.err || echo "pmc pass failed"
find $OUT -name "*kernel_stats.csv" | head -3
