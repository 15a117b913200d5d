#!/bin/bash
# cProfile of the bench's eager train step: where the HOST spends its time (the step is ~400 library calls + ~200 torch ops).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/hostprof; mkdir -p $O
cd $R && python -m cProfile -o $O/p.prof bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_kernel_events --no_fp32 --no_graph --no_ragged --no_micro > $O/b.json 2> $O/b.err
python - <<PY
import pstats
p = pstats.Stats("$O/p.prof")
p.sort_stats("tottime").print_stats(28)
PY
