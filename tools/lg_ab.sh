#!/bin/bash
# blocked path at the C5 shard: products of the polynomial route as assembly against lg_gemm_kernel
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-lg}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_asm -- python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_prof_asm.log 2>&1 &&
python3 bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_asm.json 2> gpurun_out/${tag}_bench_asm.err &&
GRAPE_LG_ASM=0 python3 bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_cpp.json 2> gpurun_out/${tag}_bench_cpp.err
rc=$?
f=$(find gpurun_out/${tag}_prof_asm -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_asm_kernel_stats.csv && head -9 "$f" | cut -c1-160
python3 -c "
import json
for t in ('asm','cpp'):
    d=json.load(open('gpurun_out/${tag}_bench_%s.json'%t)); print(t, d['value'], d['phases_ms'])
"
exit $rc
