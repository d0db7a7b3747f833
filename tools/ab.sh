# same-box A/B of one environment knob over the headline step: bash tools/ab.sh VAR "v1 v2 ..." [reps] [extra bench flags]
VAR=$1; VALS=$2; REPS=${3:-2}; EXTRA=${4:-}
B="python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-fp32-leg --no-ddp-train --no-roofline --no-other-paths $EXTRA"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "steps/s  mean", round(d["ms_per_step"],4), "median", round(d["ms_per_step_median"],4), "min", round(d["ms_per_step_min"],4))'
for rep in $(seq $REPS); do for v in $VALS; do echo -n "$VAR=$v: "; env $VAR=$v $B 2>/dev/null | python3 -c "$P"; done; done
