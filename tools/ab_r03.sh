B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-fp32-leg --no-ddp-train --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],4))'
for rep in 1 2; do
for v in "new 8" "new 16" "new 0" "oldhalo 8" "oldhalo 16"; do
  set -- $v
  lib=$PWD/3d-latent-diffusion-model_amd/libldm3d.so; [ $1 = oldhalo ] && lib=$PWD/3d-latent-diffusion-model_amd/libldm3d_oldhalo.so
  echo -n "halo=$1 fin_nb=$2: "; LDM3D_LIB=$lib LDM_FIN_NB=$2 $B 2>/dev/null | python -c "$P"
done; done
