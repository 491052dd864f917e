cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc6 && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc6
cd $GRAFT_REPO_ROOT
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/pmc6 -o pmc -- python3 scratch/flash_pmc.py > gpurun_out/pmc6/run.log 2>&1
tail -3 gpurun_out/pmc6/run.log
f=$(find gpurun_out/pmc6 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "flash" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:50], r["Counter_Name"], r["Counter_Value"], "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
