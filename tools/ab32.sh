# same-box A/B of one environment knob over the fp32 precision mode of the headline step: bash tools/ab32.sh VAR "v1 v2 ..." [reps]
VAR=$1; VALS=$2; REPS=${3:-2}
B="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-ddp-train --no-roofline --no-other-paths"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["fp32_mode"]["steps_per_s"],1), "steps/s fp32;", round(d["value"],1), "bf16")'
for rep in $(seq $REPS); do for v in $VALS; do echo -n "$VAR=$v: "; env $VAR=$v $B 2>/dev/null | python3 -c "$P"; done; done
