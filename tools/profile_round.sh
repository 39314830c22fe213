#!/bin/bash
# Bench lines and rocprofv3 kernel statistics of one build: tools/profile_round.sh <tag>
#   gpurun_out/<tag>_bench_{C3,C2,C5}.json, gpurun_out/<tag>_{C3,C5}_kernel_stats.csv
# (the summaries that are to be judged are copied to profiles/ and committed by hand)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-r04}
python3 bench.py > gpurun_out/${tag}_bench_C3.json 2> gpurun_out/${tag}_bench_C3.err &&
python3 bench.py --config C2 --steps 50 --warmup 5 > gpurun_out/${tag}_bench_C2.json 2> gpurun_out/${tag}_bench_C2.err &&
python3 bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_C5.json 2> gpurun_out/${tag}_bench_C5.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_C3 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_prof_C3.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_C5 -- python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_prof_C5.log 2>&1
rc=$?
for c in C3 C5; do
  f=$(find gpurun_out/${tag}_prof_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${c}_kernel_stats.csv
done

# secondary lines: six controls and non-Hermitian generators at the headline shape
python3 bench.py --config C3L6 --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_C3_L6.json 2> gpurun_out/${tag}_bench_C3_L6.err
python3 bench.py --config C3 --nonhermitian --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_C3_nonherm.json 2> gpurun_out/${tag}_bench_C3_nonherm.err
# general control operators as well (the streamed all-tiles derivative kernel), and its compiled twin
python3 bench.py --config C3 --nonhermitian --nonhermitian-controls --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_C3_nonherm_ctrl.json 2> gpurun_out/${tag}_bench_C3_nonherm_ctrl.err
GRAPE_DERIV3G=0 python3 bench.py --config C3 --nonhermitian --nonhermitian-controls --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_C3_nonherm_ctrl_compiled.json 2> gpurun_out/${tag}_bench_C3_nonherm_ctrl_compiled.err
# the cliff of the four-product route: every cell beyond its range (dt = 1.5: five products) and with one squaring (dt = 2)
python3 bench.py --config C3 --dt 1.5 --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_C3_dt1p5.json 2> gpurun_out/${tag}_bench_C3_dt1p5.err
python3 bench.py --config C3 --dt 2.0 --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_C3_dt2.json 2> gpurun_out/${tag}_bench_C3_dt2.err
# control operators per trajectory (the ensemble of a robustness problem), Hermitian and general
python3 bench.py --config C3 --per-trajectory-controls --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_C3_pertraj.json 2> gpurun_out/${tag}_bench_C3_pertraj.err
python3 bench.py --config C3 --per-trajectory-controls --nonhermitian --steps 8 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_bench_C3_pertraj_nonherm.json 2> gpurun_out/${tag}_bench_C3_pertraj_nonherm.err
exit $rc
