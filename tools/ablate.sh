#!/bin/bash
# Diagnostic only: builds a -DGRAPE_DIAG copy of the library, prints in-kernel s_memtime shares of
# the expm kernel and times it with parts switched off (results are wrong in ablated runs; only the
# phase times are read).  The product library is restored afterwards.
#   ABLATE (default "1 2"): expm ablations, bit0 = no tile inversions, bit1 = no solve at all
#   ABLATE_D2 (default "1"): deriv2_kernel ablations, bit0 = no parking traffic
set -e
cd "$(dirname "$0")/.."
cp grape.jl_amd/csrc/libgrape_hip.so /tmp/libgrape_hip.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DGRAPE_DIAG grape.jl_amd/csrc/grape_hip.hip -o grape.jl_amd/csrc/libgrape_hip.so
echo "== stamps"; GRAPE_DIAG_STAMPS=1 python tools/time_c3.py C3 2>&1 | grep -E "stamp|eval|solve:|cell [ab]:" | tail -26
for ab in ${ABLATE-1 2}; do echo "== expm ablate=$ab"; GRAPE_DIAG_ABLATE=$ab python tools/time_c3.py C3 2>&1 | tail -1; done
for ab in ${ABLATE_D2-1}; do echo "== deriv2 ablate=$ab"; GRAPE_DIAG_ABLATE_D2=$ab python tools/time_c3.py C3 2>&1 | tail -1; done
cp /tmp/libgrape_hip.prod.so grape.jl_amd/csrc/libgrape_hip.so
