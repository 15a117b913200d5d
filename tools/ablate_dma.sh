#!/bin/bash
# Ablation of conv_gemm_dma_kernel on the GPU box: rebuilds conv.hip with -DSMT_ABL=<mask> and times the
# k=9 / dilation 27 forward conv.  Masks: 1 no activation DMA, 2 no weight re-staging, 4 no MFMA,
# 8 no activation fragment reads, 16 no weight fragment reads.  Results are NOT numerically valid.
set -e
cd "$(dirname "$0")/../speech-masters-thesis_amd/csrc"
# ablated objects go to build_abl/ and a library of their own: the product build (build/, libsmt_hip.so) is never touched
make -s && mkdir -p build_abl && cp build/*.o build_abl/
export SMT_HIP_LIB="$PWD/../smt_amd/libsmt_hip_abl.so"
for m in ${MASKS:-0 8 32 40}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -DSMT_ABL=$m -c conv.hip -o build_abl/conv.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../smt_amd/libsmt_hip_abl.so build_abl/*.o
  echo "== SMT_ABL=$m (conv_ws: 4 no MFMA loop, 8 no fragment reads, 32 no epilogue)"
  python ../../tools/bench_dma.py 2>&1 | grep "dma=1"
done
