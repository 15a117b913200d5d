#!/bin/bash
# Ablation of conv_k1_bwd_kernel on the GPU box (-DSMT_KB_ABL: 1 no data gradient, 2 no weight gradient).
set -e
cd "$(dirname "$0")/../speech-masters-thesis_amd/csrc"
# ablated objects go to build_abl/ and a library of their own: the product build (build/, libsmt_hip.so) is never touched
make -s && mkdir -p build_abl && cp build/*.o build_abl/
export SMT_HIP_LIB="$PWD/../smt_amd/libsmt_hip_abl.so"
for m in ${MASKS:-0 1 2 3}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -DSMT_KB_ABL=$m -c conv_k1_bwd.hip -o build_abl/conv_k1_bwd.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../smt_amd/libsmt_hip_abl.so build_abl/*.o
  echo "== SMT_KB_ABL=$m"; python ../../tools/bench_k1bwd.py 2>&1 | tail -1
done
