#!/bin/bash
# Separate rocprofv3 --pmc passes (never combined with tracing, see task notes) for the bench run.
# Usage: tools/pmc.sh <tag> [config] [bench args]  -> gpurun_out/pmc_<tag>_<config>_{fetch,write,mfma}/...
# then (anywhere): python3 tools/pmc_summary.py <tag> <config>  -> profiles/<tag>_pmc_summary_<config>.json
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-r01}_${2:-C3}
cfg=${2:-C3}
shift 2 2>/dev/null || shift $#
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-matrix-free "$@" > gpurun_out/pmc_${tag}_${name}.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for name in ("fetch","write","mfma"):
    files = glob.glob(f"gpurun_out/pmc_${tag}_{name}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        print(name, k, {c: (sum(v)/len(v), len(v)) for c, v in d.items()})
PY
