# per-kernel profile of a decode configuration: B=128 CTX=512 bash scratch/prof_batch.sh
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_batch && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_batch
cd $GRAFT_REPO_ROOT
export NVR_NO_EXIT=1
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_batch -o pb -- python3 scratch/batch_profile.py > gpurun_out/prof_batch/run.log 2>&1
tail -1 gpurun_out/prof_batch/run.log
find gpurun_out/prof_batch -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/trace_stats.py {} 8 2>&1 | head -${LINES_OUT:-14}
find gpurun_out/prof_batch -name "*kernel_trace.csv" -delete
