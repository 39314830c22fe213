#!/bin/bash
# blocked path at the C5 shard: round-5 changes of phase A, A/B by environment switches (tools/c5_ab.sh <tag>)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-c5ab}
run() {   # name, env...
  name=$1; shift
  env "$@" python3 bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err || return 1
  python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_${name}.json')); print('${name}', round(d['ms_per_step'],2), d['phases_ms'])"
}
run default X=1 &&
run old GRAPE_LG_SPEC=0 GRAPE_LG_SN=0 &&
run nospec GRAPE_LG_SPEC=0 &&
run nosn GRAPE_LG_SN=0 &&
run scratch24 GRAPE_LG_SCRATCH_GB=24 &&
run scratch12 GRAPE_LG_SCRATCH_GB=12 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_prof.log 2>&1
rc=$?
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv && head -12 "$f" | cut -c1-150
exit $rc
