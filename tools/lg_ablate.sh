#!/bin/bash
# timing-only variants of lg_gemm_asm (csrc/asm/gen_lg.py, GRAPE_LG_ABLATE) at the C5 shard, without rebuilding the library:
# every code object under asm_variants/lg/ is loaded through GRAPE_ASM_CO.  Results of the ablated variants are WRONG by
# construction; only phases_ms is read.    tools/lg_ablate.sh <tag> [names...]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
tag=${1:-lgab}; shift
names=${@:-$(ls asm_variants/lg/*.co | xargs -n1 basename | sed 's/\.co$//')}
for n in $names; do
  GRAPE_ASM_CO=$PWD/asm_variants/lg/$n.co timeout -k 10 300 python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-matrix-free > gpurun_out/${tag}_$n.json 2> gpurun_out/${tag}_$n.err
  python3 -c "
import json,sys
try:
    d=json.load(open('gpurun_out/${tag}_$n.json')); print('$n', round(d['ms_per_step'],2), d['phases_ms'])
except Exception as e:
    print('$n', 'FAILED', e); print(open('gpurun_out/${tag}_$n.err').read()[-600:])
"
done
