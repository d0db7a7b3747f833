#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench command plus separate PMC passes
# (one counter group per pass, --kernel-trace only, as MI355X_MICROARCH.md prescribes).  Output -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B --steps 50 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-ddp-train --no-other-paths > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $B --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fp32-leg --no-ddp-train --no-other-paths > $OUT/bench_pmc_$name.json 2> $OUT/pmc_$name.err || echo "pmc pass $name failed"
done
ls -R $OUT | head -40
