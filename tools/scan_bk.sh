#!/bin/bash
# C2 (and the README-shaped C1) against the block length of the scanned sweeps (GRAPE_SCAN16_BK); 0 = sequential sweeps
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
for cfg in C2 C1; do
for bk in 0 8 12 16 20 24 32; do
  if [ $bk = 0 ]; then export GRAPE_SCAN16=0; unset GRAPE_SCAN16_BK; else export GRAPE_SCAN16=1 GRAPE_SCAN16_BK=$bk; fi
  python3 bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-matrix-free > gpurun_out/scanbk_${cfg}_$bk.json 2> gpurun_out/scanbk_${cfg}_$bk.err
  python3 -c "
import json
try:
    d=json.load(open('gpurun_out/scanbk_${cfg}_$bk.json')); print('$cfg', $bk, round(d['ms_per_step'],4), d['phases_ms'])
except Exception as e:
    print('$cfg', $bk, 'FAILED'); print(open('gpurun_out/scanbk_${cfg}_$bk.err').read()[-300:])
"
done; done
