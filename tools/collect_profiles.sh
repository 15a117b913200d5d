#!/bin/bash
# Runs on the GPU box (gpurun): default bench line, rocprofv3 kernel stats, and the two PMC passes.
# Outputs land in gpurun_out/$TAG/ (TAG defaults to r03); copy the summaries into profiles/ afterwards.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[1/5] default bench (with cpu_baseline)"; date
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -2 $O/bench_default.err
echo "[2/5] rocprofv3 kernel stats"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_fp32 --no_graph --no_ragged > $O/stats_bench.json 2> $O/stats_bench.err
rm -f $O/stats/*kernel_trace.csv
echo "[3/5] PMC FETCH_SIZE"; date
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_events --no_fp32 --no_graph --no_ragged > $O/fetch.json 2> $O/fetch.err
echo "[4/5] PMC WRITE_SIZE"; date
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_events --no_fp32 --no_graph --no_ragged > $O/write.json 2> $O/write.err
python $R/tools/pmc_traffic.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $O/pmc_traffic
rm -f $O/fetch/*kernel_trace.csv $O/write/*kernel_trace.csv
echo "[5/5] PMC MFMA busy + clock"; date
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o m -- python $R/bench.py --steps 1 --warmup 3 --no_cpu_baseline --no_kernel_events --no_fp32 --no_graph --no_ragged --no_micro > $O/mfma.json 2> $O/mfma.err
python $R/tools/pmc_mfma.py $O/mfma/m_counter_collection.csv $O/mfma/m_kernel_trace.csv $O/pmc_mfma
rm -f $O/mfma/*kernel_trace.csv $O/mfma/m_counter_collection.csv
echo "[6/6] TransformerLM workload: bench line + kernel stats"; date
python $R/bench.py --workload transformer_lm --steps 20 --warmup 5 > $O/lm_bench.json 2> $O/lm_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lm_stats -o lm -- python $R/bench.py --workload transformer_lm --steps 10 --warmup 3 --no_cpu_baseline --no_kernel_events > $O/lm_stats_bench.json 2> $O/lm_stats_bench.err
rm -f $O/lm_stats/*kernel_trace.csv
python $R/bench.py --workload transformer_lm --lm_graph --steps 30 --warmup 5 --no_cpu_baseline > $O/lm_bench_graph.json 2> $O/lm_bench_graph.err
python $R/bench.py --workload transformer_lm --lm_graph --lm_tune_gemm --steps 30 --warmup 5 --no_cpu_baseline > $O/lm_bench_graph_tuned.json 2> $O/lm_bench_graph_tuned.err
echo "[7/7] auxiliary paths (encode-only pass, STFT.inverse, maximum_path)"; date
python $R/bench.py --workload aux --steps 10 --warmup 3 > $O/aux_bench.json 2> $O/aux_bench.err
echo "[8/8] GlowTTS workload: bench line + kernel stats; spectral kernels stand-alone"; date
python $R/bench.py --workload glow_tts --steps 10 --warmup 3 > $O/glow_bench.json 2> $O/glow_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/glow_stats -o glow -- python $R/bench.py --workload glow_tts --steps 5 --warmup 2 --no_cpu_baseline --no_kernel_events > $O/glow_stats_bench.json 2> $O/glow_stats_bench.err
rm -f $O/glow_stats/*kernel_trace.csv
python $R/tools/bench_spectral.py > $O/spectral_micro.txt 2>&1
ls $O $O/stats $O/lm_stats $O/glow_stats
