set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1 && echo TESTS_OK && \
python bench.py > gpurun_out/r01_v5_bench.json 2> gpurun_out/r01_v5_bench.err && echo BENCH_OK && \
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > gpurun_out/final_smoke.log 2>&1 && echo SMOKE_OK && \
cd /tmp && export TMPDIR=/tmp && \
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v5 -o v5 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/v5_prof_bench.json 2> $R/gpurun_out/v5_prof.err && echo PROF_OK && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f5 -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/pmc_f5.log 2>&1 && echo PMCF_OK && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w5 -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/pmc_w5.log 2>&1 && echo PMCW_OK
