# rocprofv3 kernel stats of float32 bs-1 decode (scratch/f32_step.py)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_f32
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 scratch/f32_step.py > $out/run.log 2>&1
tail -3 $out/run.log
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
for name, v in rows[:12]:
    print(f'{sum(v)/1e3:9.2f} ms  n={len(v):6d}  median {statistics.median(v):8.2f}  min {min(v):8.2f}  max {max(v):8.2f} us  {name[:90]}')
PY
find $out -name "*.csv" -delete
