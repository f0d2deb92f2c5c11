# Measurement sequence, one gpurun call: rocprofv3 kernel stats, the two PMC passes, a bench run (the launch counts the PMC
# summary needs), the summary, and the bench line again - now carrying roofline.traffic from that summary (tools/pmc_traffic.py, which also records the source fingerprint the bench line checks).  usage:
#   bash tools/r2_measure.sh <tag> [bench.py flags of the workload ...]
set -o pipefail
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
cd /tmp && export TMPDIR=/tmp && \
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o p -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof.err && echo PROF_OK && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f_${TAG} -o f -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/${TAG}_pmc_f.log 2>&1 && echo PMCF_OK && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w_${TAG} -o w -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/${TAG}_pmc_w.log 2>&1 && echo PMCW_OK && \
cd $R && python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && echo BENCH_OK && \
python3 tools/pmc_traffic.py gpurun_out/pmc_f_${TAG}/f_counter_collection.csv gpurun_out/pmc_w_${TAG}/w_counter_collection.csv gpurun_out/${TAG}_traffic_pmc.json gpurun_out/${TAG}_bench.json && \
cp gpurun_out/${TAG}_traffic_pmc.json profiles/${TAG}_traffic_pmc.json && \
python bench.py "$@" > gpurun_out/${TAG}_bench.json 2>> gpurun_out/${TAG}_bench.err && echo BENCH2_OK && \
cp gpurun_out/prof_${TAG}/p_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv && \
rm -rf gpurun_out/pmc_f_${TAG} gpurun_out/pmc_w_${TAG} gpurun_out/prof_${TAG}/p_kernel_trace.csv && echo SUMMARY_OK
