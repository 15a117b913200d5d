#!/bin/bash
# PMC pass for the issue-bound question: vector-ALU instructions and busy cycles per kernel of one train step.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${TAG:-r03}_valu; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/v -o v -- python $R/bench.py --steps 1 --warmup 2 --no_cpu_baseline --no_kernel_events --no_fp32 --no_graph --no_ragged --no_micro > $O/v.json 2> $O/v.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/l -o l -- python $R/bench.py --steps 1 --warmup 2 --no_cpu_baseline --no_kernel_events --no_fp32 --no_graph --no_ragged --no_micro > $O/l.json 2> $O/l.err || true
python $R/tools/pmc_valu.py $O/v/v_counter_collection.csv $O/l/l_counter_collection.csv > $O/pmc_valu.txt
rm -f $O/v/*kernel_trace.csv $O/l/*kernel_trace.csv
cat $O/pmc_valu.txt
