#!/bin/bash
# Diagnostic build of conv.hip with -DSMT_WS_STAMP=1 (per-wave cycle sums of the phases of conv_ws2_kernel) into a library of
# its own (the product build is never touched), then tools/ws_phases.py.  Run on the GPU box.
set -e
cd "$(dirname "$0")/../speech-masters-thesis_amd/csrc"
make -s && mkdir -p build_abl && cp build/*.o build_abl/
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -DSMT_WS_STAMP=1 -c conv.hip -o build_abl/conv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../smt_amd/libsmt_hip_abl.so build_abl/*.o
export SMT_HIP_LIB="$PWD/../smt_amd/libsmt_hip_abl.so"
cd ../.. && python3 tools/ws_phases.py
