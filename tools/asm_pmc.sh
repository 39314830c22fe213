#!/bin/bash
# wave-cycle counters of assembly-kernel variants through the stand-alone harness: tools/asm_pmc.sh <tag> a.co [b.co ...]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=$1; shift
for pass in "a:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
            "b:SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 tools/asm_bench.py --reps 2 "$@" > gpurun_out/pmc_${tag}_${name}.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_${tag}_${name}.log; }
done
python3 - <<PY
import csv, glob, collections
for name in ("a", "b"):
    files = glob.glob(f"gpurun_out/pmc_${tag}_{name}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            if "expm_t16_asm" in row["Kernel_Name"]:
                agg[row["Dispatch_Id"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    # dispatches come in groups of (1 + reps) per variant
    ids = sorted(agg, key=int)
    for n, i in enumerate(ids):
        m = {c: sum(v) for c, v in agg[i].items()}
        wc = m.get("SQ_WAVE_CYCLES", 1)
        print(name, "dispatch", n, " ".join(f"{c[3:] if c.startswith('SQ_') else c}={m[c]:.4g}({100 * m[c] / wc:.1f}%)" for c in sorted(m)))
PY
