# rocprofv3 kernel medians of the decode steps of scratch/route_scan.py at ONE batch size: bash scratch/prof_scan.sh qwen3-0.6b 256 64
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_scan
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT; export NVR_NO_EXIT=1
rocprofv3 --kernel-trace --output-format csv -d $out -o r -- python3 scratch/route_scan.py $1 $2 $3 > $out/run.log 2>&1
tail -1 $out/run.log
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "argmax_partials" in r["Kernel_Name"]]
rows = rows[ends[-16] + 1:ends[-1] + 1]
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
print(f"last 15 decode steps: {tot/15:.1f} us of kernel time per step")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {sum(v)/15:8.1f} us/step  n/step {len(v)/15:5.1f}  median {statistics.median(v):7.2f}  {k[:110]}")
PY
find $out -name "*kernel_trace.csv" -delete
