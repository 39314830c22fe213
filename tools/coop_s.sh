#!/bin/bash
# cooperative sweeps of the blocked path at the C5 shard: siblings per trajectory, per direction (round 5; tools/coop_s.sh <tag>)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-coops}
for sf in 32 16 8 4; do for sb in 32 16 8; do
  GRAPE_COOP_S_FW=$sf GRAPE_COOP_S_BW=$sb timeout -k 10 120 python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/${tag}.json')); print('fw S=$sf bw S=$sb', d['phases_ms']['forward'], d['phases_ms']['backward'])"
done; done
