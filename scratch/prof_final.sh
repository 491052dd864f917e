cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_final && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_final
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o r01final -- python3 bench.py > gpurun_out/prof_final/bench.log 2>&1
tail -1 gpurun_out/prof_final/bench.log > gpurun_out/prof_final/bench_line_under_rocprof.json
find gpurun_out/prof_final -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/trace_stats.py {} 32 > gpurun_out/prof_final/breakdown.txt 2>&1
find gpurun_out/prof_final -name "*kernel_trace.csv" -delete
python3 bench.py > gpurun_out/prof_final/bench_line.json 2> gpurun_out/prof_final/bench_stderr.log
tail -1 gpurun_out/prof_final/bench_line.json | cut -c1-400
head -14 gpurun_out/prof_final/breakdown.txt
