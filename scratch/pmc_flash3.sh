cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc11 && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc11
cd $GRAFT_REPO_ROOT
export REPS=2
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU"; do
i=$((i+1))
timeout 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc11/s$i -o pmc -- python3 scratch/$1.py ${3:-} > gpurun_out/pmc11/s$i.log 2>&1
f=$(find gpurun_out/pmc11/s$i -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" "$2" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
d = list(agg.values())[-1]
print("  ".join(f"{k}={v:.4g}" for k, v in d.items()))
PY
done
