# same-box A/B of one environment knob over the AutoencoderKL 96^3 encode / decode: bash tools/ab_vae.sh VAR "v1 v2 ..." [reps] [dims]
VAR=$1; VALS=$2; REPS=${3:-2}; DIMS=${4:-96,96,96}
for rep in $(seq $REPS); do for v in $VALS; do echo -n "$VAR=$v: "; env $VAR=$v python3 tools/bench_vae.py $DIMS 2>/dev/null | tail -1; done; done
