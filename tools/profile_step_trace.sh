set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04f_step
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-ddp-train --no-other-paths --no-roofline > $OUT/bench.txt 2> $OUT/err.txt
find $OUT -name "*kernel_trace.csv" | head -2
