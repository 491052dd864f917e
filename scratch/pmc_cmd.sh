cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc5 && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc5
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/pmc5 -o pmc -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc5/bench.log 2>&1
f=$(find gpurun_out/pmc5 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
    n = len(d.get("FETCH_SIZE", []))
    f = sum(d.get("FETCH_SIZE", [0])) / max(n, 1); w = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [])), 1)
    print(f"{k:62s} n={n:5d}  FETCH_SIZE {f:12.0f} KiB/launch  WRITE_SIZE {w:12.0f} KiB/launch")
PY
find gpurun_out/pmc5 -name "*.csv" -size +2M -delete
