# r06: counters of the LM head over 512 rows on the 256 x 256 tiles (scratch/lm_head_bench.py 512): matrix-pipe busy share and HBM bytes per launch.
# Separate rocprofv3 --pmc passes, --kernel-trace only beside them (MI355X_MICROARCH.md).  usage (GPU box): bash tools/pmc_lm_head.sh
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_lmhead
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/busy -o pmc -- python3 scratch/lm_head_bench.py 512 > $out/busy.log 2>&1; echo "busy rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o pmc -- python3 scratch/lm_head_bench.py 512 > $out/fetch.log 2>&1; echo "fetch rc=$?"
python3 - $out <<'PY'
import csv, glob, sys, collections, statistics
out = sys.argv[1]
def rows(sub):
    f = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    return [r for r in csv.DictReader(open(f)) if "gemm256_kernelILi4" in r["Kernel_Name"]]
d = collections.defaultdict(dict)
for r in rows("busy"): d[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
busy = [v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["GRBM_GUI_ACTIVE"] / 8) for v in d.values() if "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0]
fetch = [float(r["Counter_Value"]) * 2048 / 1e6 for r in rows("fetch") if r["Counter_Name"] == "FETCH_SIZE"]
print(f"gemm256_kernel<GEPI_LMHEAD>, 512 rows x 151 936 x 1024: matrix-pipe busy {statistics.median(busy):.3f} (median of {len(busy)} launches; SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8))")
print(f"HBM read per launch {statistics.median(fetch):.1f} MB (FETCH_SIZE x 2; W = 311.2 MB once, x = 1 MB): {statistics.median(fetch) / 311.2:.2f} x the weight bytes")
PY
find $out -name "*.csv" -delete
