#!/bin/bash
# PMC passes over the halo conv micro-benchmark (full kernel vs ablations) -> gpurun_out/pmc_halo/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_halo
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for abl in 0 8 4; do
  i=0
  for grp in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TD_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TCR_RDRET_STALL_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    i=$((i+1))
    LDM_CONV_DBG=$abl timeout -k 10 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/a${abl}_g$i -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py 256,256,24,24,24,2,1 > /dev/null 2> $OUT/a${abl}_g$i.err || echo "pass a$abl g$i failed"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_halo"
for abl in (0, 8, 4):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"{out}/a{abl}_g*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "conv3_halo" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("ABL", abl, {k: round(acc[k] / max(1, n[k]), 1) for k in sorted(acc)})
PY
