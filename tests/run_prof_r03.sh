# Round-3 profile collection (run on the GPU box from the repo root): kernel trace of the driver's bench command, busy/idle per step.
set -e
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/stats -o c3 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03/stats.log 2>&1
cd $R
T=$(ls gpurun_out/r03/stats/*/c3_kernel_trace.csv 2>/dev/null | head -1); [ -z "$T" ] && T=$(ls gpurun_out/r03/stats/c3_kernel_trace.csv)
python tests/trace_gaps.py $T 20 > gpurun_out/r03/c3_last20.txt
S=$(ls gpurun_out/r03/stats/*/c3_kernel_stats.csv 2>/dev/null | head -1); [ -z "$S" ] && S=$(ls gpurun_out/r03/stats/c3_kernel_stats.csv)
python tests/prof_summary.py $S 45 > gpurun_out/r03/c3_kernel_summary.txt
cat gpurun_out/r03/c3_last20.txt; tail -1 gpurun_out/r03/stats.log | cut -c1-300
