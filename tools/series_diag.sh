#!/bin/bash
# Diagnostic only: -DGRAPE_SERIES_DIAG build of the library (in-kernel s_memtime shares of the matrix-free sweep),
# one C3 evaluation, then the product library is restored.
set -e
cd "$(dirname "$0")/.."
cp grape.jl_amd/csrc/libgrape_hip.so /tmp/libgrape_hip.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DGRAPE_SERIES_DIAG grape.jl_amd/csrc/grape_hip.hip -o grape.jl_amd/csrc/libgrape_hip.so
python tools/time_series.py C3 2>&1 | grep -E "series" | tail -8
cp /tmp/libgrape_hip.prod.so grape.jl_amd/csrc/libgrape_hip.so
