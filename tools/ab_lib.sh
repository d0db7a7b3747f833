# same-box A/B of two builds of the library: bash tools/ab_lib_r03.sh libA.so libB.so [reps]
A=$1; B=$2; REPS=${3:-3}
CMD="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-fp32-leg --no-ddp-train --no-roofline --no-other-paths"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],4))'
for rep in $(seq $REPS); do for L in $A $B; do echo -n "$(basename $L): "; LDM3D_LIB=$PWD/3d-latent-diffusion-model_amd/$L $CMD 2>/dev/null | python -c "$P"; done; done
