cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc10 && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc10
cd $GRAFT_REPO_ROOT
export REPS=2
for prog in gemm_pmc flash_pmc; do
timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc10/$prog -o pmc -- python3 scratch/$prog.py > gpurun_out/pmc10/$prog.log 2>&1
echo "$prog rc=$?"
f=$(find gpurun_out/pmc10/$prog -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm256" in r["Kernel_Name"] or "flash" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:45], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for k, d in agg.items():
    print(k[0], d, "conflict/active = %.3f" % (d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
done
