#!/bin/bash
# blocked path at the C5 shard: round-5 switches of phase A, A/B inside one box (tools/c5_lanes.sh <tag>)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-c5lanes}
run() {   # name, env...
  name=$1; shift
  env "$@" timeout -k 10 120 python3 bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err || return 1
  python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_${name}.json')); print('${name}', round(d['ms_per_step'],2), d['phases_ms']['expm'])"
}
for rep in a b; do
run default_$rep X=1 &&
run lanes2_$rep GRAPE_LG_LANES=2 &&
run form1_$rep GRAPE_LG_FORM2=0 &&
run nofuse_$rep GRAPE_LG_FUSE=0 &&
run r05start_$rep GRAPE_LG_FUSE=0 GRAPE_LG_FORM2=0 || exit 1
done
