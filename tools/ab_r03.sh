# same-box A/B of one knob: bash tools/ab_r03.sh VAR "v1 v2 ..." [reps]
VAR=$1; VALS=$2; REPS=${3:-2}
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-fp32-leg --no-ddp-train --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],4))'
for rep in $(seq $REPS); do for v in $VALS; do echo -n "$VAR=$v: "; env $VAR=$v $B 2>/dev/null | python -c "$P"; done; done
