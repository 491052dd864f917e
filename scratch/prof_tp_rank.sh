# per-kernel profile of ONE tensor-parallel rank's decode step (compute only, collectives skipped): bash scratch/prof_tp_rank.sh <tp>
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tp && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_tp
cd $GRAFT_REPO_ROOT
export NVR_NO_EXIT=1
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tp -o tp -- python3 scratch/tp_rank_compute.py $1 > gpurun_out/prof_tp/run.log 2>&1
tail -1 gpurun_out/prof_tp/run.log
find gpurun_out/prof_tp -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/trace_stats.py {} 16 2>&1 | head -${LINES_OUT:-14}
find gpurun_out/prof_tp -name "*kernel_trace.csv" -delete
