#!/bin/bash
# Runs on the GPU box (gpurun): default bench line, rocprofv3 kernel stats, and the two PMC passes.
# Outputs land in gpurun_out/r01b/; copy the summaries into profiles/ afterwards.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r01b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[1/4] default bench (with cpu_baseline)"; date
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -2 $O/bench_default.err
echo "[2/4] rocprofv3 kernel stats"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline > $O/stats_bench.json 2> $O/stats_bench.err
rm -f $O/stats/*kernel_trace.csv
echo "[3/4] PMC FETCH_SIZE"; date
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_events > $O/fetch.json 2> $O/fetch.err
echo "[4/4] PMC WRITE_SIZE"; date
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_events > $O/write.json 2> $O/write.err
python $R/tools/pmc_traffic.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $O/pmc_traffic
rm -f $O/fetch/*kernel_trace.csv $O/write/*kernel_trace.csv
ls $O $O/stats
