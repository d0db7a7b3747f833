set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04e_vae_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/bench_train_vae.py --steps 10 > $OUT/bench.txt 2> $OUT/err.txt
find $OUT -name "*kernel_stats.csv" | head -2
