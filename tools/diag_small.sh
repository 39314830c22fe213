set -e
cd $GRAFT_REPO_ROOT
cp grape.jl_amd/csrc/libgrape_hip.so /tmp/libgrape_hip.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DGRAPE_DIAG grape.jl_amd/csrc/grape_hip.hip -o grape.jl_amd/csrc/libgrape_hip.so
GRAPE_DIAG_STAMPS=1 python tools/time_c3.py C2 128 2>&1 | grep -E "stamp|eval" | tail -16
cp /tmp/libgrape_hip.prod.so grape.jl_amd/csrc/libgrape_hip.so
