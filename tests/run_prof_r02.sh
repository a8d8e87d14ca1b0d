set -e
mkdir -p gpurun_out/r02
timeout -k 10 400 python bench.py > gpurun_out/r02/bench_c3.json 2> gpurun_out/r02/bench_c3.err
timeout -k 10 200 python bench.py --workload c4 --no-cpu-baseline > gpurun_out/r02/bench_c4.json 2> gpurun_out/r02/bench_c4.err
cd /tmp && export TMPDIR=/tmp && export MI_PHYSICS_NO_GRAPH=1
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/stats -o c3 -- python3 $R/bench.py --steps 100 --no-cpu-baseline > $R/gpurun_out/r02/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02/pmc_fetch -o c3 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r02/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02/pmc_write -o c3 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r02/pmc_write.log 2>&1
cd $R; ls -la gpurun_out/r02/*; cat gpurun_out/r02/bench_c3.json gpurun_out/r02/bench_c4.json
