# rocprofv3 kernel stats of a python script: bash tools/prof_script.sh <tag> <script.py> [ENV=...]
tag=$1; script=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $script > $out/run.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*.csv" ! -name kernel_stats.csv -delete
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["Percentage"]):5.1f} %')
PY
