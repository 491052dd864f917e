# rocprofv3 kernel stats of scratch/bench_8b.py (Qwen3-8B shapes, one GPU); usage: bash tools/prof_8b.sh [env assignments...]
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_8b
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
export NVR_NO_EXIT=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o q8b -- python3 scratch/bench_8b.py > $out/run.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
find $out -name "*.csv" ! -name kernel_stats.csv -delete
grep "tp=" $out/run.log
head -14 $out/kernel_stats.csv | cut -c1-170
