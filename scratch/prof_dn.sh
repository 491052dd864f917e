cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_dn && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_dn
cd $GRAFT_REPO_ROOT
for DN in 1 0; do
export NVR_DEFERRED_NORM=$DN
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dn -o dn$DN -- python3 bench.py --steps 60 --warmup 10 --eager --no-cpu-baseline > gpurun_out/prof_dn/bench$DN.log 2>&1
find gpurun_out/prof_dn -name "dn${DN}*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/trace_stats.py {} 32 > gpurun_out/prof_dn/breakdown$DN.txt 2>&1
find gpurun_out/prof_dn -name "*kernel_trace.csv" -delete
head -12 gpurun_out/prof_dn/breakdown$DN.txt
done
