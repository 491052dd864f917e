cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_bs512
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bs512 -- python3 scratch/bs512.py > $out/run.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
find $out -name "*.csv" ! -name kernel_stats.csv -delete
cat $out/run.log | tail -3
head -14 $out/kernel_stats.csv | cut -c1-170
