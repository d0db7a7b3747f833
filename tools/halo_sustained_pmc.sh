#!/bin/bash
# MFMA-busy share of conv3_halo_kernel per 24^3 shape.  Runs on the GPU box (gpurun): one rocprofv3 --pmc pass per
# shape over tools/halo_sustained.py --iters N (kernel-trace + counters only, as MI355X_MICROARCH.md prescribes) -> gpurun_out/halo_pmc/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/halo_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in plain wide skip; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/$shape -- python3 $GRAFT_REPO_ROOT/tools/halo_sustained.py --shapes $shape --iters 4000 > $OUT/$shape.txt 2> $OUT/$shape.err || echo "pmc pass $shape failed"
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/halo_pmc"
for shape in ("plain", "wide", "skip"):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"{out}/{shape}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "conv3_halo" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    dur = []
    for f in glob.glob(f"{out}/{shape}/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "conv3_halo" in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    avg = {k: acc[k] / max(1, n[k]) for k in acc}
    d_us = sum(dur) / max(1, len(dur)) / 1e3
    line = f"{shape}: {len(dur)} launches, avg {d_us:.2f} us (serialised by the counter collection); per launch: " + ", ".join(f"{k} {v:.4g}" for k, v in sorted(avg.items()))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CYCLES" in avg:
        line += f" | MFMA busy per SIMD = {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.0f} cycles"
    print(line)
PY
