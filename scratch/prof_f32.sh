# rocprofv3 kernel stats of float32 bs-1 decode (scratch/f32_step.py)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_f32
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 scratch/f32_step.py > $out/run.log 2>&1
tail -3 $out/run.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms  n={int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Name"][:110]}')
PY
find $out -name "*.csv" -delete
