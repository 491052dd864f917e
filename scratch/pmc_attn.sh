cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc7 && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc7
cd $GRAFT_REPO_ROOT
export LAYERS=4 CTX=1044
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc7 -o pmc -- python3 scratch/attn_tune.py "" > gpurun_out/pmc7/run.log 2>&1
tail -2 gpurun_out/pmc7/run.log
f=$(find gpurun_out/pmc7 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, json
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "attn_rows" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
vals = [float(r["Counter_Value"]) for r in rows]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(rows), rows[0]["Kernel_Name"][:90])
print("FETCH_SIZE KiB min/med/max", min(vals), sorted(vals)[len(vals)//2], max(vals), "dur_us med", sorted(durs)[len(durs)//2])
open("gpurun_out/pmc7/summary.json", "w").write(json.dumps({"kernel": rows[0]["Kernel_Name"], "n": len(rows), "fetch_size_kib_median": sorted(vals)[len(vals)//2], "fetch_size_kib_min": min(vals), "fetch_size_kib_max": max(vals), "duration_us_under_pmc_median": sorted(durs)[len(durs)//2]}))
PY
head -60 "$f" > gpurun_out/pmc7/counter_collection_head.csv
find gpurun_out/pmc7 -name "*.csv" -size +1M -delete
