#!/bin/bash
# Ablation of vq_search_kernel on the GPU box: rebuilds vq.hip with -DVQ_ABL=<mask> and times the 36,352 x 1024 search.
# Masks: 1 no MFMAs, 2 no re-staging of the codebook, 4 no best / runner-up folding, 8 no epilogue, 32 no fragment reads.
# Results are NOT valid.  Phase timestamps: build with -DVQ_ABL=16 and run tools/vq_phases.py.
set -e
cd "$(dirname "$0")/../speech-masters-thesis_amd/csrc"
# ablated objects go to build_abl/ and a library of their own: the product build (build/, libsmt_hip.so) is never touched
make -s && mkdir -p build_abl && cp build/*.o build_abl/
export SMT_HIP_LIB="$PWD/../smt_amd/libsmt_hip_abl.so"
for m in ${MASKS:-0 1 2 4 8 15}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -DVQ_ABL=$m -c vq.hip -o build_abl/vq.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../smt_amd/libsmt_hip_abl.so build_abl/*.o
  echo "== VQ_ABL=$m"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$m -o vq -- python3 $OLDPWD/../../tools/bench_vq.py 2 > /tmp/abl$m.log 2>&1; python3 $OLDPWD/../../tools/kstats.py /tmp/abl$m/vq_kernel_stats.csv 12 | grep "search\|candid\|exact\|reduce")
done
