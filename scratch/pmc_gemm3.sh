# counter passes over tools/prefill_layer_bench.py (one pass per set; --kernel-trace only): L2 hit rate and HBM traffic of the prefill GEMM variants
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm3
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
i=$((i+1))
timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/s$i -o pmc -- python3 tools/prefill_layer_bench.py > $out/s$i.log 2>&1
f=$(find $out/s$i -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm256" in k or "flash" in k or "rmsnorm" in k:
        agg[k[:60] + "|" + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, "  ".join(f"{c}={sorted(v)[len(v)//2]:.4g} (n={len(v)})" for c, v in d.items()))
PY
find $out/s$i -name "*.csv" -delete
done
