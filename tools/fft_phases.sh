#!/bin/bash
# Diagnostic build of spectral.hip with -DSMT_FFT_STAMP=1 (cycle sums of the phases of the one-wave spectral-loss forward
# kernel) into a library of its own (the product build is never touched), then tools/fft_phases.py.  Run on the GPU box.
set -e
cd "$(dirname "$0")/../speech-masters-thesis_amd/csrc"
make -s && mkdir -p build_abl && cp build/*.o build_abl/
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -DSMT_FFT_STAMP=1 -c spectral.hip -o build_abl/spectral.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../smt_amd/libsmt_hip_abl.so build_abl/*.o
export SMT_HIP_LIB="$PWD/../smt_amd/libsmt_hip_abl.so"
cd ../.. && python3 tools/fft_phases.py
