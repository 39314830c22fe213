#!/bin/bash
# Wave-cycle breakdown of the kernels of one bench run (separate --pmc pass, never combined with tracing):
#   tools/pmc_sq.sh <tag> [bench args]   ->  gpurun_out/pmc_<tag>_sq/ + a per-kernel table on stdout
# WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stalls) + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-t18}; shift
# PMC=... selects another set of at most 8 SQ counters (keep SQ_WAVE_CYCLES in it: the table is relative to it)
PMC=${PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES}
rocprofv3 --pmc $PMC \
  --output-format csv -d gpurun_out/pmc_${tag}_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matrix-free "$@" > gpurun_out/pmc_${tag}_sq.log 2>&1
python3 - <<PY
import csv, glob, collections
files = glob.glob("gpurun_out/pmc_${tag}_sq/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc < 1e6: continue
    print(k, "launches", len(d["SQ_WAVE_CYCLES"]))
    for c in sorted(m): print("   %-28s %14.4g  %6.1f %% of wave cycles" % (c, m[c], 100 * m[c] / wc))
PY
