# rocprofv3 kernel trace of configs[4] decode (scratch/bs512.py): per-kernel medians over the decode steps
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_bs512
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT; export NVR_NO_EXIT=1
rocprofv3 --kernel-trace --output-format csv -d $out -o r -- python3 scratch/bs512.py > $out/run.log 2>&1
tail -2 $out/run.log
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 20 decode steps: between the 21st-last and the last end-of-step argmax launch
ends = [i for i, r in enumerate(rows) if "argmax_partials" in r["Kernel_Name"]]
rows = rows[ends[-21] + 1:ends[-1] + 1]
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
print(f"last 20 decode steps: {len(rows)} launches, {tot/20:.1f} us of kernel time per step, {tot/20/28:.2f} us per layer")
for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(f'{sum(v)/20:9.1f} us/step {100*sum(v)/tot:5.1f}%  n/step={len(v)/20:6.1f}  median {statistics.median(v):8.2f}  min {min(v):8.2f} us  {name[:120]}')
PY
find $out -name "*.csv" -delete
