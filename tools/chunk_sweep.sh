#!/bin/bash
# blocked path at the C5 shard against the size of a chunk of cells (GRAPE_LG_CHUNK): the nine scratch arrays of a chunk
# are 9.4 MB per cell -- a chunk of <= ~24 cells lives in the 256 MB Infinity Cache, so the intermediates of the five
# products would never reach HBM; the price is launches of a few hundred workgroups.   tools/chunk_sweep.sh <tag> [sizes...]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-chunk}; shift
sizes=${@:-"635 256 128 64 32 24 16 8"}
for n in $sizes; do
  GRAPE_LG_CHUNK=$n timeout -k 10 300 python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_$n.json 2> gpurun_out/${tag}_$n.err
  python3 -c "
import json
try:
    d=json.load(open('gpurun_out/${tag}_$n.json')); print($n, round(d['ms_per_step'],2), d['phases_ms'])
except Exception as e:
    print($n, 'FAILED', e); print(open('gpurun_out/${tag}_$n.err').read()[-400:])
"
done
