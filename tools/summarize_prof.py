"""Summarise a tools/profile_round.sh output directory into profiles/<tag>_summary.{md,json}.

HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
counts 128-B requests as 64 B for wide coalesced streams, so the read side is DOUBLED before use; WRITE_SIZE is exact
for 16-B-per-lane stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01a"
commit = sys.argv[2] if len(sys.argv) > 2 else None          # the commit the profiled library was built from (bench.py reports it)
src = os.path.join("gpurun_out", f"prof_{tag}")
out = {"commit": commit}


def short(n):
    return n.split("(")[0].replace("void ", "")[:60]


# ---- kernel-trace stats -------------------------------------------------------------------------------------------
f = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
nsteps = bench["warmup"] + bench["steps"] * (2 if bench.get("roofline") else 1)   # roofline leg repeats the K steps
out["bench_under_profiler"] = {k: bench[k] for k in ("value", "ms_per_step", "steps", "warmup")}
out["bench_roofline_hip_events"] = bench.get("roofline")
out["kernels"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                   "per_step_ms": float(r["TotalDurationNs"]) / 1e6 / nsteps, "pct": 100 * float(r["TotalDurationNs"]) / tot}
                  for r in rows[:14]]
out["gpu_busy_ms_per_step"] = tot / 1e6 / nsteps

# ---- PMC passes: per-kernel-name sums / dispatch ---------------------------------------------------------------------
pmc = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k][r["Counter_Name"]] += 1
dom = next((k for k in pmc if k.startswith("conv3_halo_kernel")), "conv_igemm_kernel<2, 2, 64, 4, 2>")
per = {c: pmc[dom][c] / max(1, ndisp[dom][c]) for c in pmc[dom]}
out["conv_pmc_per_launch"] = per
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    out["conv_hbm_traffic_bytes_per_launch"] = {
        "kernel": dom, "launches": int(ndisp[dom]["FETCH_SIZE"]),
        "read_corrected": per["FETCH_SIZE"] * 1024 * 2, "write": per["WRITE_SIZE"] * 1024,
        "total": per["FETCH_SIZE"] * 1024 * 2 + per["WRITE_SIZE"] * 1024,
        "note": "FETCH_SIZE x1024 x2 (gfx950 half-count correction) + WRITE_SIZE x1024, averaged over all launches of the kernel"}
if "SQ_VALU_MFMA_BUSY_CYCLES" in per and "SQ_BUSY_CYCLES" in per and per["SQ_BUSY_CYCLES"]:
    out["conv_mfma_busy_over_sq_busy"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / per["SQ_BUSY_CYCLES"]
if "SQ_LDS_BANK_CONFLICT" in per and per.get("SQ_LDS_IDX_ACTIVE"):
    out["conv_lds_bank_conflict_frac"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
other = {}
for k in pmc:
    if k != dom and "FETCH_SIZE" in pmc[k]:
        other[k] = {"fetch_kib_per_launch": pmc[k]["FETCH_SIZE"] / ndisp[k]["FETCH_SIZE"],
                    "write_kib_per_launch": pmc[k].get("WRITE_SIZE", 0) / max(1, ndisp[k].get("WRITE_SIZE", 1))}
out["other_kernels_hbm_kib"] = other

os.makedirs("profiles", exist_ok=True)
json.dump(out, open(os.path.join("profiles", f"{tag}_summary.json"), "w"), indent=1)
with open(os.path.join("profiles", f"{tag}_summary.md"), "w") as w:
    w.write(f"# rocprofv3 summary {tag}\n\ncommand: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline`\n\n")
    w.write(f"bench under the profiler: {bench['value']:.1f} steps/s ({bench['ms_per_step']:.3f} ms/step); GPU busy {out['gpu_busy_ms_per_step']:.3f} ms/step\n\n")
    w.write("| kernel | calls | avg us | ms/step | % |\n|---|---|---|---|---|\n")
    for k in out["kernels"]:
        w.write(f"| {k['name']} | {k['calls']} | {k['avg_us']:.2f} | {k['per_step_ms']:.3f} | {k['pct']:.1f} |\n")
    w.write("\nPMC (separate passes, per launch of the dominant conv kernel):\n\n```\n" + json.dumps(
        {k: out[k] for k in out if k.startswith("conv_")}, indent=1) + "\n```\n")
print(json.dumps({k: out[k] for k in out if k != "other_kernels_hbm_kib"}, indent=1)[:3500])
