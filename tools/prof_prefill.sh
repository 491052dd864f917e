# rocprofv3 kernel stats of ONE engine prefill step (scratch/prefill_step.py): bash tools/prof_prefill.sh <tag> [NVR_LIBNVR=...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_prefill_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 scratch/prefill_step.py > $out/run.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*.csv" ! -name kernel_stats.csv -delete
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}')
PY
