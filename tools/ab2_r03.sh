B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-fp32-leg --no-ddp-train --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],4))'
for rep in 1 2; do for v in "0 1" "1 1" "0 0" "1 0"; do set -- $v; echo -n "XCD_ROWS=$1 WT=$2: "; LDM_XCD_ROWS=$1 LDM_WT_STORES=$2 $B 2>/dev/null | python -c "$P"; done; done
