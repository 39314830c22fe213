"""Condense the rocprofv3 --pmc passes of tools/pmc.sh <tag> <config> into profiles/<tag>_pmc_summary_<config>.json
(bench.py reads the summary of ITS configuration, or reports null).

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are KB per dispatch, FETCH_SIZE is doubled on gfx950; GRBM_GUI_ACTIVE is summed over the 8 XCDs;
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 4 SIMDs * 256 CUs)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag0 = sys.argv[1] if len(sys.argv) > 1 else "r01"
cfg = sys.argv[2] if len(sys.argv) > 2 else "C3"
tag = f"{tag0}_{cfg}"
out = collections.defaultdict(dict)
for name in ("fetch", "write", "mfma"):
    # gpurun merges every call's files into gpurun_out/: only the newest run of a pass is the current build
    files = sorted(glob.glob(f"gpurun_out/pmc_{tag}_{name}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        shutil.copy(f, f"profiles/{tag0}_pmc_{name}_counter_collection_{cfg}.csv")
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            out[k][c] = sum(v) / len(v)
            out[k]["dispatches"] = len(v)
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    if d.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
        d["executed_mfma_flop_per_launch"] = 512.0 * d["SQ_INSTS_VALU_MFMA_MOPS_F64"]
    if d.get("GRBM_GUI_ACTIVE"):
        d["MfmaUtil_percent"] = 100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
json.dump({"source": f"tools/pmc.sh {tag} (rocprofv3 --pmc, separate passes FETCH_SIZE | WRITE_SIZE | SQ/GRBM), "
                     f"bench.py --config {cfg} --steps 3 --warmup 1; condensed by tools/pmc_summary.py",
           "note": "means per dispatch; FETCH_SIZE/WRITE_SIZE in KB, FETCH doubled per the gfx950 correction; "
                   "GRBM_GUI_ACTIVE is summed over 8 XCDs",
           "kernels": out}, open(f"profiles/{tag0}_pmc_summary_{cfg}.json", "w"), indent=1)
for k, d in out.items():
    print(k[:60], {c: round(v, 1) for c, v in d.items() if c in ("hbm_bytes_per_launch", "MfmaUtil_percent", "dispatches")})
