cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof8b && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof8b
cd $GRAFT_REPO_ROOT
export NVR_NO_EXIT=1
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof8b -o p8b -- python3 scratch/bench_8b.py > gpurun_out/prof8b/run.log 2>&1
tail -1 gpurun_out/prof8b/run.log
find gpurun_out/prof8b -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/trace_stats.py {} 16 2>&1 | head -${LINES_OUT:-14}
find gpurun_out/prof8b -name "*kernel_trace.csv" -delete
