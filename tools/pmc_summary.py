#!/usr/bin/env python
"""Condense gpurun_out/prof_<tag>/ (written on the GPU box by tools/profile.sh) into committed summaries under
profiles/: <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the HEADLINE command), <tag>_secondary_
kernel_stats.csv (side measurements), <tag>_pmc_summary.{md,json} (per-kernel PMC averages from the separate --pmc
passes) and profiles/pmc_latest.json (read by bench.py for roofline.traffic). Kernels of the headline loop take their
counters from the headline passes only; the others (encoder, streaming scan, loss) from the side-measurement passes."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
OUT = sys.argv[2] if len(sys.argv) > 2 else "profiles"  # on the GPU box: a directory under gpurun_out/ (raw traces exceed 64 MiB)
os.makedirs(OUT, exist_ok=True)
shutil.copy(f"{src}/trace/t_kernel_stats.csv", f"{OUT}/{tag}_kernel_stats.csv")
if os.path.exists(f"{src}/sec_trace/t_kernel_stats.csv"):
    shutil.copy(f"{src}/sec_trace/t_kernel_stats.csv", f"{OUT}/{tag}_secondary_kernel_stats.csv")


def collect(prefix):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(os.listdir(src)):
        f = f"{src}/{d}/p_counter_collection.csv"
        if not d.startswith(prefix) or not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": len(next(iter(cs.values())))}
            for k, cs in acc.items() if k.startswith("t2l::")}


head = collect("pmc_")
sec = {k: v for k, v in collect("sec_").items() if k not in head}
summary = {**{k: dict(v, source="headline command") for k, v in head.items()},
           **{k: dict(v, source="side measurements") for k, v in sec.items()}}
for k, s in summary.items():
    if "FETCH_SIZE" in s:
        # rocprofv3 FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
        # (MI355X_MICROARCH.md §HBM; re-checked here: split_db_kernel reads 1 KiB/row and reports 0.5) -> doubled;
        # WRITE_SIZE taken as is (uncalibrated).
        s["hbm_bytes_per_launch"] = 2 * s["FETCH_SIZE"] * 1024 + s.get("WRITE_SIZE", 0.0) * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in s and s.get("GRBM_GUI_ACTIVE", 0) > 0:
        cyc = s["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        s["kernel_cycles"] = cyc
        s["mfma_pipe_busy_frac"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc  # 1024 SIMDs
json.dump(summary, open(f"{OUT}/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
json.dump(summary, open(f"{OUT}/pmc_latest.json", "w"), indent=1, sort_keys=True)
with open(f"{OUT}/{tag}_pmc_summary.md", "w") as f:
    f.write(f"# PMC summary `{tag}` — rocprofv3 --pmc passes (separate runs; see tools/profile.sh)\n\n")
    for k, s in sorted(summary.items()):
        f.write(f"## {k}  ({s['source']})\n\n| counter | average per launch |\n|---|---|\n")
        for c, v in sorted(s.items()):
            if c != "source":
                f.write(f"| {c} | {v:,.3f} |\n")
        f.write("\n")
for k, s in sorted(summary.items()):
    print(k, s["source"], {c: round(v, 2) for c, v in s.items() if c in ("hbm_bytes_per_launch", "mfma_pipe_busy_frac", "kernel_cycles", "launches")})
