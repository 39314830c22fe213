#!/bin/bash
# Phase costs of expm_t18_kernel by truncation: builds the kernel's translation unit with -DT18_STOP=n (the cell is left
# after phase n: 1 form A, 2 A2, 3 A3, 4 planes <- A3, 5 A6, 6 exchanges + norms, 7 combinations + planes <- B1, 8 A9,
# 9 planes <- B3 + A9; no T18_STOP: the whole cell) and links it against the product objects.
#   here:       tools/t18_ablate.sh build      ->  tools/_stop<n>.so
#   GPU box:    tools/t18_ablate.sh run [cfg]  ->  phase-A time of every variant
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
C=grape.jl_amd/csrc
if [ "$1" = build ]; then
  for n in 1 2 3 4 5 6 7 8 9; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -DT18_STOP=$n -c $C/grape_t18.hip -o /tmp/t18_stop$n.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC $C/grape_hip.o /tmp/t18_stop$n.o -o tools/_stop$n.so || exit 1
  done
else
  cfg=${2:-C3}
  for n in 1 2 3 4 5 6 7 8 9; do
    python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import api, synth
api._lib = None
api.library_path = lambda: os.path.abspath("tools/_stop$n.so")
pr = synth.make_config("$cfg")
h = g.GrapeHip(pr['H0'], pr['Hc'], pr['tlist'], pr['psi0'], pr['target'], pr['weights'])
for it in range(2):
    try: h.eval(pr['pulsevals'])
    except Exception: pass
h.reset_timings()
for it in range(5):
    try: h.eval(pr['pulsevals'])
    except Exception: pass
print("stop $n: expm %.3f ms" % h.timings()['expm'], flush=True)
PY
  done
fi
