#!/bin/bash
# derivative kernel: assembly against its compiled twin at the headline shape (kernel statistics + bench lines)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-d3}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_asm -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_prof_asm.log 2>&1 &&
GRAPE_DERIV3_ASM=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_cpp -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_prof_cpp.log 2>&1 &&
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_asm.json 2> gpurun_out/${tag}_bench_asm.err &&
GRAPE_DERIV3_ASM=0 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_cpp.json 2> gpurun_out/${tag}_bench_cpp.err
rc=$?
for c in asm cpp; do
  f=$(find gpurun_out/${tag}_prof_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${c}_kernel_stats.csv && head -8 "$f"
done
cat gpurun_out/${tag}_bench_asm.json gpurun_out/${tag}_bench_cpp.json
exit $rc
