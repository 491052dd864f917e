cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc8 && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc8
cd $GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc8 -o pmc -- python3 scratch/gemm_pmc.py > gpurun_out/pmc8/run.log 2>&1
echo "rc=$?"; tail -2 gpurun_out/pmc8/run.log
f=$(find gpurun_out/pmc8 -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm256" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:40], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"], "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
