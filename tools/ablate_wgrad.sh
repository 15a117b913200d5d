#!/bin/bash
# Ablation of conv_wgrad_dma_kernel on the GPU box: rebuilds conv_wgrad.hip with -DSMT_WABL=<mask>.
# Masks: 1 stage the operand tiles once only, 2 no fragment reads, 4 no MFMA.  Results are NOT numerically valid.
set -e
cd "$(dirname "$0")/../speech-masters-thesis_amd/csrc"
# ablated objects go to build_abl/ and a library of their own: the product build (build/, libsmt_hip.so) is never touched
make -s && mkdir -p build_abl && cp build/*.o build_abl/
export SMT_HIP_LIB="$PWD/../smt_amd/libsmt_hip_abl.so"
for m in ${MASKS:-0 1 2 4 6 7}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -DSMT_WABL=$m -c conv_wgrad.hip -o build_abl/conv_wgrad.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../smt_amd/libsmt_hip_abl.so build_abl/*.o
  echo "== SMT_WABL=$m"
  python ../../tools/bench_wgrad.py 2>&1 | grep "wgrad k"
done
