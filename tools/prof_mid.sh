# rocprofv3 kernel stats of scratch/mid_batch.py <shape>: bash tools/prof_mid.sh <tag> <BxCTX> [NVR_LIBNVR=...]
tag=$1; shape=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_mid_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 scratch/mid_batch.py $shape > $out/run.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*.csv" ! -name kernel_stats.csv -delete
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["Percentage"]):5.1f} %')
PY
