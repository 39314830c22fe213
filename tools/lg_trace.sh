#!/bin/bash
# per-dispatch durations of the blocked path at the C5 shard, grouped by kernel and grid size (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-lgtrace}
rm -rf gpurun_out/${tag}_prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}.log 2>&1
f=$(find gpurun_out/${tag}_prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"][:28]
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    agg[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
tot = sum(sum(v) for v in agg.values())
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 0.002 * tot: continue
    print(f"{name:30s} grid {grid:9d} x{len(v):5d}  avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}  total {sum(v)*1e-3:8.2f} ms")
PY
