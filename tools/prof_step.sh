# rocprofv3 kernel trace of bench.py (decode step breakdown); usage: bash tools/prof_step.sh <tag> [env assignments...]
# run on the GPU box through gpurun; results under gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shared-prefix --no-configs3 --no-prefill-sweep --no-batch-sweep --no-default-engine --no-live-pmc > $out/bench.log 2>&1
grep '^{' $out/bench.log | tail -1 > $out/bench_line_under_rocprof.json   # the JSON line, not rocprofv3's last log line
find $out -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/trace_stats.py {} 16 > $out/breakdown.txt 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
find $out -name "*.csv" ! -name kernel_stats.csv -delete
head -40 $out/breakdown.txt
