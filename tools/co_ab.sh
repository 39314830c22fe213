#!/bin/bash
# A/B/A/B of whole code objects (GRAPE_ASM_CO) on one configuration: tools/co_ab.sh <config> <dir> name1 name2 [bench args]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
cfg=$1; dir=$2; a=$3; b=$4; shift 4
for v in $a $b $a $b; do
  GRAPE_ASM_CO=$PWD/$dir/$v.co python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-matrix-free "$@" > gpurun_out/coab_$v.json 2> gpurun_out/coab_$v.err
  python3 -c "
import json
d=json.load(open('gpurun_out/coab_$v.json')); print('$v', round(d['ms_per_step'],3), d['phases_ms'])"
done
