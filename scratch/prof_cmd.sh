cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof4 && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof4
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -o r01v4 -- python3 bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/prof4/bench.log 2>&1
tail -1 gpurun_out/prof4/bench.log | cut -c1-200
find gpurun_out/prof4 -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/trace_stats.py {} 16 > gpurun_out/prof4/breakdown.txt 2>&1
head -30 gpurun_out/prof4/breakdown.txt
find gpurun_out/prof4 -name "*kernel_trace.csv" -delete
ls -la gpurun_out/prof4 gpurun_out/prof4/*
