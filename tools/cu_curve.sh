#!/bin/bash
# How the phases of the C3 evaluation scale with the number of CUs their stream may use (GRAPE_STREAM_CUS, a diagnostic
# switch of grape_create): the sweeps are HBM-bound, the derivative kernel is matrix-bound -- the curve says what a
# side-by-side schedule of the two on disjoint CU sets could gain (round-5 review item 5).   tools/cu_curve.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-cucurve}
for n in 256 224 192 160 128 96 64; do
  GRAPE_STREAM_CUS=$n GRAPE_GRAPH=0 timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_$n.json 2> gpurun_out/${tag}_$n.err
  python3 -c "
import json
try:
    d=json.load(open('gpurun_out/${tag}_$n.json')); print($n, round(d['ms_per_step'],3), d['phases_ms'])
except Exception as e:
    print($n, 'FAILED', e); print(open('gpurun_out/${tag}_$n.err').read()[-400:])
"
done
