# Round-3 profile collection (run on the GPU box from the repo root).  Kernel trace + PMC passes of the DRIVER's bench command
# (bench.py --steps 20 --warmup 5), a kernel trace of the default window, config 4, and the bench lines themselves.
set -e
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_c3_driver.json 2> gpurun_out/r03/bench_c3_driver.err
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_c3_default.json 2> gpurun_out/r03/bench_c3_default.err
timeout -k 10 200 python bench.py --workload c4 --no-cpu-baseline > gpurun_out/r03/bench_c4.json 2> gpurun_out/r03/bench_c4.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/stats -o c3 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/stats_c4 -o c4 -- python3 $R/bench.py --workload c4 --no-cpu-baseline > $R/gpurun_out/r03/stats_c4.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r03/pmc_fetch -o c3 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r03/pmc_write -o c3 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/r03/pmc_sq -o c3 -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/r03/pmc_tcc -o c3 -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03/pmc_tcc.log 2>&1
cd $R
(python tests/pmc_summary.py gpurun_out/r03/pmc_sq/c3_counter_collection.csv 10; python tests/pmc_summary.py gpurun_out/r03/pmc_tcc/c3_counter_collection.csv 10) > gpurun_out/r03/pmc_stage_kernels.txt 2>&1 || true
python tests/trace_gaps.py gpurun_out/r03/stats/c3_kernel_trace.csv 20 > gpurun_out/r03/c3_last20.txt
python tests/prof_summary.py gpurun_out/r03/stats/c3_kernel_stats.csv 45 > gpurun_out/r03/c3_kernel_summary.txt
python tests/prof_summary.py gpurun_out/r03/stats_c4/c4_kernel_stats.csv 25 > gpurun_out/r03/c4_kernel_summary.txt
(python tests/pmc_summary.py gpurun_out/r03/pmc_fetch/c3_counter_collection.csv 20; python tests/pmc_summary.py gpurun_out/r03/pmc_write/c3_counter_collection.csv 20) 2>&1 | grep -E "kernel|k_cl_solve|k_epa|k_pairs" > gpurun_out/r03/pmc_summary.txt || true
rm -rf gpurun_out/r03/pmc_sq gpurun_out/r03/pmc_tcc gpurun_out/r03/pmc_fetch/*.db gpurun_out/r03/pmc_write/*.db; find gpurun_out/r03 -name '*_kernel_trace.csv' -size +20M -delete
cat gpurun_out/r03/c3_last20.txt | head -5; cat gpurun_out/r03/pmc_summary.txt; tail -c 600 gpurun_out/r03/bench_c3_driver.json
