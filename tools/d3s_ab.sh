#!/bin/bash
# six controls at the headline shape: streamed-controls assembly kernel against deriv2_kernel's STREAM_L form
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-d3s}
python3 bench.py --config C3L6 --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_asm.json 2> gpurun_out/${tag}_bench_asm.err &&
GRAPE_DERIV3S=0 python3 bench.py --config C3L6 --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_cpp.json 2> gpurun_out/${tag}_bench_cpp.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_asm -- python3 bench.py --config C3L6 --steps 4 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_prof_asm.log 2>&1
rc=$?
f=$(find gpurun_out/${tag}_prof_asm -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_asm_kernel_stats.csv && head -6 "$f"
python3 -c "
import json
for t in ('asm','cpp'):
    d=json.load(open('gpurun_out/${tag}_bench_%s.json'%t)); print(t, d['value'], d['phases_ms'], d.get('deriv_kernel'))
"
exit $rc
